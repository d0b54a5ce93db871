#!/usr/bin/env python3
"""Launch one bf16 GEMM configuration a few times (for rocprofv3 --pmc runs).
usage: gemm_probe.py nt|tn M N K shape [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402

kind, M, N, K, shape = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(bf)  # noqa: E731
vitk.lib.vitk_debug_force_gemm_shape(shape)
if kind == "nt":
    a, w = r(M, K), r(N, K)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    for _ in range(iters):
        ops.gemm_nt(a, w, out, vitk.EPI_NONE)
else:
    dy, x = r(K, M), r(K, N)      # reduction length K (tokens)
    c = torch.zeros(M, N, device="cuda")
    for _ in range(iters):
        ops.gemm_tn_acc(dy, x, c, 0)
torch.cuda.synchronize()
