#!/usr/bin/env python3
"""Error of the fused MLP-hidden backward du = (dz W2) * dropout * gelu'(u) against fp64, for two builds of the
kernel library in one process (the in-tree one and tools/bin/libvitk_old.so, which keeps u in bf16 and
evaluates gelu' in the backward epilogue)."""
import ctypes
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    for name, args in vitk.PROTOTYPES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = args, ctypes.c_int
    lib.vitk_error_string.restype = ctypes.c_char_p
    lib.vitk_error_string.argtypes = [ctypes.c_int]
    return lib


old = load(os.path.join(ROOT, "tools", "bin", "libvitk_old.so"))
bf = torch.bfloat16
for (T, D, F, sx) in [(60, 64, 128, 1.0), (60, 64, 128, 0.2), (4096, 768, 3072, 1.0), (4096, 768, 3072, 3.0)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(T, D, device="cuda", generator=g) * sx).to(bf)
    w1 = (torch.randn(F, D, device="cuda", generator=g) / math.sqrt(D)).to(bf)
    b1 = torch.randn(F, device="cuda", generator=g) * 0.1
    dz = torch.randn(T, D, device="cuda", generator=g).to(bf)
    w2t = (torch.randn(F, D, device="cuda", generator=g) / math.sqrt(F)).to(bf)     # [F, D] = W2^T
    u64 = x.double() @ w1.double().t() + b1.double()
    u64.requires_grad_(True)
    torch.nn.functional.gelu(u64).backward(dz.double() @ w2t.double().t())
    ref = u64.grad
    for name, lib in (("new", vitk.lib), ("old", old)):
        ops.lib = lib
        aux = torch.empty(T, F, dtype=bf, device="cuda")
        gg = torch.empty(T, F, dtype=bf, device="cuda")
        du = torch.empty(T, F, dtype=bf, device="cuda")
        ops.gemm_nt(x, w1, gg, vitk.EPI_BIAS_GELU, bias=b1, aux=aux)
        ops.gemm_nt(dz, w2t, du, vitk.EPI_DGELU, aux=aux)
        err = (du.double() - ref)
        print(f"T={T} F={F} |x|~{sx}: {name}: max err / max {float(err.abs().max() / ref.abs().max()):.3e}, "
              f"rms err / rms {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}")
    ops.lib = vitk.lib
