#!/usr/bin/env python3
"""Where the cycles of the weight-gradient GEMM loop go: the instrumented library (make -C vit-pytorch_amd/csrc abl)
with parts of the steady-state loop compiled out (vitk_debug_tn_abl: 1 = no LDS-DMA, 2 = no fragment reads, 4 = no
barriers; results garbage, times valid), the write-out skipped throughout, same process, interleaved.
usage: VITK_LIB=tools/bin/libvitk_abl.so python tools/tn_abl.py [batch]"""
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
for fn in (vitk.lib.vitk_debug_tn_abl, vitk.lib.vitk_debug_tn_no_flush):
    fn.argtypes, fn.restype = [ctypes.c_int], None
MASKS = [0, 1, 2, 3, 4, 7]
vitk.lib.vitk_debug_tn_no_flush(1)
for name, M, N in [("dW qkv", 3 * D, D), ("dW fc1", F, D), ("dW fc2", D, F)]:
    dy, x = r(T, M), r(T, N)
    c = torch.zeros(M, N, device="cuda")
    res = {}
    for rnd in range(5):
        for m in MASKS:
            vitk.lib.vitk_debug_tn_abl(m)
            ops.gemm_tn_acc(dy, x, c, 0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm_tn_acc(dy, x, c, 0)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(m, []).append(e0.elapsed_time(e1) / 8)
    fl = 2.0 * T * M * N
    print(f"{name} [{M}x{N}] T={T}: " + "  ".join(f"abl{m}: {statistics.median(v):.3f} ms ({fl / statistics.median(v) / 1e9:.0f})" for m, v in res.items()))
vitk.lib.vitk_debug_tn_abl(0)
vitk.lib.vitk_debug_tn_no_flush(0)
