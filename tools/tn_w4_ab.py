#!/usr/bin/env python3
"""Weight-gradient GEMM: the four-wave kernel (one wave per SIMD, 128 x 128 per wave) against the eight-wave
ping-pong kernel, same process, alternating; results compared first (the products must be the same bits, the
column sums equal to rounding).  usage: tn_w4_ab.py [batch]"""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T, D, F = B * 197, 768, 3072
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(bf)  # noqa: E731
vitk.lib.vitk_debug_tn_w4.argtypes = [ctypes.c_int]
vitk.lib.vitk_debug_tn_w4.restype = None
ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
assert vitk.lib.vitk_gemm_tn_set_workspace(ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel())) == 0
cases = [("dW qkv", 3 * D, D), ("dW proj", D, D), ("dW fc1", F, D), ("dW fc2", D, F), ("two cols", D, 512), ("ragged", 1000, 520), ("odd", 520, 1000)]
res = {}
for name, M, N in cases:
    Tc = T if name not in ("ragged", "odd") else 32 * 2001
    dy, x = r(Tc, M), r(Tc, N)
    outs = []
    for w4 in (0, 1):
        vitk.lib.vitk_debug_tn_w4(w4)
        c = torch.zeros(M, N, device="cuda")
        db = torch.zeros(M, device="cuda")
        ops.gemm_tn_acc(dy, x, c, 0, colsum=db)
        c2 = torch.zeros(M, N, device="cuda")
        ops.gemm_tn_acc(dy, x, c2, 0)
        torch.cuda.synchronize()
        outs.append((c, db, c2))
    ref_db = dy.float().sum(0)
    print(f"{name:8s} product same bits: {torch.equal(outs[0][0], outs[1][0])} / without column sums: {torch.equal(outs[0][2], outs[1][2])}; "
          f"column sums vs fp32 sum: eight-wave {float((outs[0][1] - ref_db).abs().max() / ref_db.abs().max()):.2e}, "
          f"four-wave {float((outs[1][1] - ref_db).abs().max() / ref_db.abs().max()):.2e}", flush=True)
    c = torch.zeros(M, N, device="cuda")
    db = torch.zeros(M, device="cuda")
    for rnd in range(5):
        for w4 in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            vitk.lib.vitk_debug_tn_w4(w4)
            ops.gemm_tn_acc(dy, x, c, 0, colsum=db)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm_tn_acc(dy, x, c, 0, colsum=db)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((name, w4), []).append(e0.elapsed_time(e1) / 8)
    a, b = statistics.median(res[(name, 0)]), statistics.median(res[(name, 1)])
    fl = 2.0 * Tc * M * N
    print(f"{name:8s} [{M}x{N}] T={Tc}: eight-wave {a:.3f} ms ({fl / a / 1e9:.0f} TF/s), four-wave {b:.3f} ms ({fl / b / 1e9:.0f} TF/s): "
          f"{100 * (b - a) / a:+.1f} %", flush=True)
vitk.lib.vitk_debug_tn_w4(0)
