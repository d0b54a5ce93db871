#!/usr/bin/env python3
"""Loop ablations of the four-wave weight-gradient kernel (times only; the ablated variants compute garbage)."""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402
T, M, N = 512 * 197, 3072, 768
bf = torch.bfloat16
dy = (torch.randn(T, M, device="cuda") * 0.5).to(bf)
x = (torch.randn(T, N, device="cuda") * 0.5).to(bf)
vitk.lib.vitk_debug_tn_w4.argtypes = [ctypes.c_int]
vitk.lib.vitk_debug_tn_w4.restype = None
vitk.lib.vitk_debug_tn_no_flush.argtypes = [ctypes.c_int]
vitk.lib.vitk_debug_tn_no_flush.restype = None
vitk.lib.vitk_debug_tn_no_flush(1)
c = torch.zeros(M, N, device="cuda")
names = {0: "eight-wave", 1: "four-wave", 2: "four-wave, no DMA", 3: "four-wave, no reads", 4: "four-wave, neither", 8: "four-wave, neither, no barriers"}
res = {}
for rnd in range(4):
    for mode in names:
        vitk.lib.vitk_debug_tn_w4(mode)
        ops.gemm_tn_acc(dy, x, c, 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ops.gemm_tn_acc(dy, x, c, 0)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) / 8)
vitk.lib.vitk_debug_tn_w4(0)
vitk.lib.vitk_debug_tn_no_flush(0)
for mode, nm in names.items():
    t = statistics.median(res[mode])
    print(f"{nm:34s} {t:.3f} ms  {2.0 * T * M * N / t / 1e9:.0f} TF/s")
