#!/usr/bin/env python3
"""Is the weight-gradient GEMM waiting on memory?  The same launches with every token row aliased to row 0
(row stride 0: all operand reads hit the caches) against the real operands, same process, interleaved."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402,F401
from vitk import ops  # noqa: E402

vitk.lib.vitk_debug_tn_w4.argtypes = [ctypes.c_int]
vitk.lib.vitk_debug_tn_w4.restype = None
vitk.lib.vitk_debug_tn_w4(0)          # this probe is about the eight-wave kernel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T, D, F = B * 197, 768, 3072
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(bf)  # noqa: E731
cases = [("dW qkv", 3 * D, D), ("dW proj", D, D), ("dW fc1", F, D), ("dW fc2", D, F)]
for name, M, N in cases:
    dy, x = r(T, M), r(T, N)
    forms = {"real": (dy, x), "aliased": (dy[:1].expand(T, M), x[:1].expand(T, N)),
             "32-row window": (dy[:32].unsqueeze(0).expand(T // 32, 32, M).reshape(T, M) if False else dy, x)}
    forms.pop("32-row window")
    c = torch.zeros(M, N, device="cuda")
    db = torch.zeros(M, device="cuda")
    res = {}
    for rnd in range(5):
        for k, (a, b) in forms.items():
            ops.gemm_tn_acc(a, b, c, 0, colsum=db)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm_tn_acc(a, b, c, 0, colsum=db)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(k, []).append(e0.elapsed_time(e1) / 8)
    fl = 2.0 * T * M * N
    print(f"{name:8s} [{M}x{N}] T={T}: " + ", ".join(f"{k} {statistics.median(v):.3f} ms ({fl / statistics.median(v) / 1e9:.0f} TF/s)" for k, v in res.items()))
