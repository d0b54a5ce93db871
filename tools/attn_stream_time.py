#!/usr/bin/env python3
"""Per-kernel times of the streaming attention path (sequences longer than 224 tokens) through the public operator:
   python tools/attn_stream_time.py [B H N]   (default 64 16 577: ViT-L/16 at 384 px, batch 64)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
from vitk import ops  # noqa: E402

B, H, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 16, 577)
D = 64 * H
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).bfloat16().requires_grad_(True)
dout = torch.randn(B * N, D, device="cuda").bfloat16()


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = [None]
fwd = timed(lambda: out.__setitem__(0, ops.attention(qkv, B, N, H)[0]))
bwd = timed(lambda: torch.autograd.grad(out[0], qkv, dout, retain_graph=True))
fl = 4.0 * B * H * N * N * 64
print(f"B={B} H={H} N={N}: forward {fwd * 1e3:.0f} us ({fl / fwd / 1e9:.0f} TF/s), backward {bwd * 1e3:.0f} us ({2.5 * fl / bwd / 1e9:.0f} TF/s)")
