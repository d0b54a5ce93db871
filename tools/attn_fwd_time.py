#!/usr/bin/env python3
"""Time the bf16 attention forward (training form: writes O, its fp32 copy and LSE).  usage: attn_fwd_time.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import torch  # noqa: E402
import vitk  # noqa: E402,F401
from vitk import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
oversub = int(sys.argv[2]) if len(sys.argv) > 2 else 1
vitk.check(vitk.lib.vitk_attn_set_oversubscription(oversub))
N, H, D = 197, 12, 768
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_(True)
ts = []
for it in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out, _ = ops.attention(qkv, B, N, H)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"attention fwd B={B} oversubscription {oversub}: {min(ts[2:]) * 1e3:.1f} us (incl. output allocation)")
