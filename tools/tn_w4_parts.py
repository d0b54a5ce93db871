#!/usr/bin/env python3
"""Four-wave vs eight-wave weight-gradient kernel with and without the column sums and the write-out."""
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
for f in ("vitk_debug_tn_w4", "vitk_debug_tn_no_flush"):
    getattr(vitk.lib, f).argtypes = [ctypes.c_int]
    getattr(vitk.lib, f).restype = None
ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
assert vitk.lib.vitk_gemm_tn_set_workspace(ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel())) == 0
c = torch.zeros(M, N, device="cuda")
db = torch.zeros(M, device="cuda")
res = {}
for rnd in range(4):
    for w4 in (0, 1):
        for cs in (0, 1):
            for nf in (1, 0):
                vitk.lib.vitk_debug_tn_w4(w4)
                vitk.lib.vitk_debug_tn_no_flush(nf)
                kw = dict(colsum=db) if cs else {}
                ops.gemm_tn_acc(dy, x, c, 0, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    ops.gemm_tn_acc(dy, x, c, 0, **kw)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((w4, cs, nf), []).append(e0.elapsed_time(e1) / 8)
vitk.lib.vitk_debug_tn_w4(0)
vitk.lib.vitk_debug_tn_no_flush(0)
for k, v in res.items():
    t = statistics.median(v)
    print(f"{'four' if k[0] else 'eight'}-wave colsum={k[1]} write-out={1 - k[2]}: {t:.3f} ms  {2.0 * T * M * N / t / 1e9:.0f} TF/s")
