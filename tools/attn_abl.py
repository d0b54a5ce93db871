#!/usr/bin/env python3
"""Time the fused attention backward with parts of the kernel compiled out (ablation build).

Build the instrumented library first: make -C vit-pytorch_amd/csrc abl   (-> tools/bin/libvitk_abl.so)
usage: attn_abl.py [B]
mask bits (feeder-wave kernel): 1 no dQ product, 2 no delta/LSE pass, 4 no exp/dS arithmetic,
8 no dV/dK products, 16 no S/dP products, 32 no DMA wait, 64 no DMA beyond the prologue (stale tiles),
128 compute waves idle; 67 = compute waves alone, 131 = DMA and its waits alone.
Forward (VITK_ATTN_FWD_ABL): 1 no K/V load, 2 one key tile only, 4 no stores."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VITK_LIB"] = os.path.join(ROOT, "tools", "bin", "libvitk_abl.so")
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import torch  # noqa: E402
import vitk  # noqa: E402
from vitk import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, H, D = 197, 12, 768
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_(True)
dout = (torch.randn(B * N, D, device="cuda") * 0.5).to(torch.bfloat16)
vitk.lib.vitk_debug_force_attn_streaming(2)        # the forward switches live in the one-shot LDS-resident kernel
for abl in (0, 1, 2, 3, 4, 5, 6, 7, 0):
    os.environ["VITK_ATTN_FWD_ABL"] = str(abl)
    ts = []
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out, _ = ops.attention(qkv, B, N, H)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"fwd abl {abl:2d}: {min(ts[1:]) * 1e3:8.1f} us", flush=True)
os.environ["VITK_ATTN_FWD_ABL"] = "0"
vitk.lib.vitk_debug_force_attn_streaming(0)
for abl in (0, 1, 2, 3, 64, 67, 128, 131, 0):
    os.environ["VITK_ATTN_ABL"] = str(abl)
    ts = []
    for it in range(6):
        out, _ = ops.attention(qkv, B, N, H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out.backward(dout)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        qkv.grad = None
    print(f"abl {abl:2d}: bwd {min(ts[1:]) * 1e3:8.1f} us", flush=True)
