#!/usr/bin/env python3
"""How much of the weight-gradient GEMM is its atomic write-out?  Same process, interleaved: the ping-pong
TN kernel with and without the final atomics (vitk_debug_tn_no_flush; the second gives wrong results)."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402

vitk.lib.vitk_debug_tn_w4.argtypes = [ctypes.c_int]
vitk.lib.vitk_debug_tn_w4.restype = None
vitk.lib.vitk_debug_tn_w4(0)          # this probe is about the eight-wave kernel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T, D, F = B * 197, 768, 3072
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(bf)  # noqa: E731
vitk.lib.vitk_debug_tn_no_flush.argtypes = [ctypes.c_int]
vitk.lib.vitk_debug_tn_no_flush.restype = None
cases = [("dW qkv", 3 * D, D), ("dW proj", D, D), ("dW fc1", F, D), ("dW fc2", D, F)]
res = {}
for name, M, N in cases:
    dy, x = r(T, M), r(T, N)
    c = torch.zeros(M, N, device="cuda")
    db = torch.zeros(M, device="cuda")
    for rnd in range(5):
        for nf in (0, 1):
            vitk.lib.vitk_debug_tn_no_flush(nf)
            ops.gemm_tn_acc(dy, x, c, 0, colsum=db)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm_tn_acc(dy, x, c, 0, colsum=db)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((name, nf), []).append(e0.elapsed_time(e1) / 8)
vitk.lib.vitk_debug_tn_no_flush(0)
for name, M, N in cases:
    a, b = statistics.median(res[(name, 0)]), statistics.median(res[(name, 1)])
    fl = 2.0 * T * M * N
    print(f"{name:8s} [{M}x{N}] T={T}: {a:.3f} ms ({fl / a / 1e9:.0f} TF/s) with the atomic write-out, {b:.3f} ms ({fl / b / 1e9:.0f} TF/s) without: "
          f"write-out {1e3 * (a - b):.0f} us = {100 * (a - b) / a:.0f} %")
