#!/usr/bin/env python3
"""A/B of the bf16 attention backward variants in one process.  usage: attn_ab.py [B] [modes...]
modes: 0 persistent with feeder wave (default), 1 streaming, 3 one-shot fused."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import torch  # noqa: E402
import vitk  # noqa: E402
from vitk import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
modes = [int(m) for m in sys.argv[2:]] or [3, 0, 3, 0]
# modes >= 10: default kernels with the persistent grid oversubscribed by (mode - 10)
N, H, D = 197, 12, 768
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_(True)
dout = (torch.randn(B * N, D, device="cuda") * 0.5).to(torch.bfloat16)
ref = None
for mode in modes:
    vitk.lib.vitk_debug_force_attn_streaming(mode if mode < 10 else 0)
    vitk.check(vitk.lib.vitk_attn_set_oversubscription(mode - 10 if mode >= 10 else 1))
    ts = []
    for it in range(6):
        out, _ = ops.attention(qkv, B, N, H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out.backward(dout)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        gr = qkv.grad.float()
        qkv.grad = None
    if ref is None:
        ref = gr
    err = ((gr - ref).abs().max() / ref.abs().max()).item()
    print(f"mode {mode}: bwd {min(ts[1:]) * 1e3:8.1f} us   max|d - d_first|/max = {err:.2e}", flush=True)
vitk.lib.vitk_debug_force_attn_streaming(0)
vitk.lib.vitk_attn_set_oversubscription(1)
