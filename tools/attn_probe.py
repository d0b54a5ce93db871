#!/usr/bin/env python3
"""Launch bf16 attention forward+backward a few times (for rocprofv3 --pmc runs).
usage: attn_probe.py [B] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402,F401
from vitk import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, H, D = 197, 12, 768
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_(True)
dout = (torch.randn(B * N, D, device="cuda") * 0.5).to(torch.bfloat16)
for _ in range(iters):
    out, _ = ops.attention(qkv, B, N, H)
    out.backward(dout)
torch.cuda.synchronize()
