#!/usr/bin/env python3
"""Per-kernel timing at the ViT-B/16 (batch 256) shapes.  Usage: python tools/kbench.py [B]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    parts = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else {"nt", "tn", "attn", "ln", "misc"}
    N, D, F, H = 197, 768, 3072, 12
    T = B * N
    bf = torch.bfloat16
    r = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(bf)  # noqa: E731
    print(f"T={T}")
    for name, (M, Nn, K) in {"qkv": (T, 3 * D, D), "out": (T, D, D), "fc1": (T, F, D), "fc2": (T, D, F)}.items():
        a, w = r(M, K), r(Nn, K)
        out = torch.empty(M, Nn, device=DEV, dtype=bf)
        bias = torch.randn(Nn, device=DEV)
        res = r(M, Nn)
        aux = torch.empty(M, Nn, device=DEV, dtype=bf)
        fl = 2.0 * M * Nn * K
        out32 = torch.empty(M, Nn, device=DEV) if Nn == D else None
        res32 = torch.randn(M, Nn, device=DEV) if Nn == D else None
        nt_shapes = tuple(int(x) for x in os.environ.get("KBENCH_NT_SHAPES", "4,5").split(","))
        for shape in nt_shapes if "nt" in parts else ():
            vitk.lib.vitk_debug_force_gemm_shape(shape)
            if out32 is not None:     # the residual-stream instance: fp32 in, fp32 out, dropout
                ms = timeit(lambda: ops.gemm_nt(a, w, out32, epi=vitk.EPI_BIAS_RES, bias=bias, residual=res32,
                                                dropout_p=0.1, seed=5))
                print(f"gemm_nt shape{shape} {name:4s} {'bias_res_f32_drop':15s} M={M} N={Nn} K={K}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF/s")
            for epi, kw in (("none", dict(epi=vitk.EPI_NONE)),
                            ("bias_res", dict(epi=vitk.EPI_BIAS_RES, bias=bias, residual=res)),
                            ("bias_gelu_drop", dict(epi=vitk.EPI_BIAS_GELU, bias=bias, aux=aux, dropout_p=0.1, seed=5)),
                            ("dgelu", dict(epi=vitk.EPI_DGELU, aux=aux))):
                ms = timeit(lambda: ops.gemm_nt(a, w, out, **kw))
                print(f"gemm_nt shape{shape} {name:4s} {epi:15s} M={M} N={Nn} K={K}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF/s")
        # weight gradient: dW[Nn,K] = dY[T,Nn]^T X[T,K]
        dy, x = r(M, Nn), r(M, K)
        c = torch.zeros(Nn, K, device=DEV)
        for shape in (0, 4) if "tn" in parts else ():
            vitk.lib.vitk_debug_force_gemm_shape(shape)
            for sk in (0,):
                ms = timeit(lambda: ops.gemm_tn_acc(dy, x, c, sk))
                print(f"gemm_tn shape{shape} {name:4s} split={sk:2d} M={Nn} N={K} T={M}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF/s")
        vitk.lib.vitk_debug_force_gemm_shape(-1)
        db = torch.zeros(Nn, device=DEV)
        ms = timeit(lambda: ops.colsum_acc(dy, db))
        print(f"colsum  {name:4s} [{M},{Nn}]: {ms:.3f} ms  {M * Nn * 2 / ms / 1e6:.0f} GB/s")
    if not parts & {"attn", "ln", "misc"}:
        return
    qkv = r(T, 3 * D)
    ms = timeit(lambda: ops.attention(qkv, B, N, H))
    fl = 4.0 * B * H * N * N * 64
    print(f"attn fwd (inference form: no fp32 copy of O): {ms:.3f} ms {fl / ms / 1e9:.0f} TF/s")
    x = qkv.clone().requires_grad_(True)
    out, _ = ops.attention(x, B, N, H)
    dout = r(T, D)
    def attn_bwd():
        x.grad = None                      # otherwise the accumulation into x.grad is timed as well
        out.backward(dout, retain_graph=True)
    ms = timeit(attn_bwd)
    print(f"attn bwd: {ms:.3f} ms {2.5 * fl / ms / 1e9:.0f} TF/s (5 products), {T * D * 18 / ms / 1e6:.0f} GB/s of its 18 B per element")
    xx = r(T, D)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    ms = timeit(lambda: ops.layer_norm(xx, g, b))
    print(f"ln fwd (bf16 in, bf16 out): {ms:.3f} ms {T * D * 4 / ms / 1e6:.0f} GB/s")
    xg = xx.clone().requires_grad_(True)
    y = ops.layer_norm(xg, g, b)
    def ln_bwd():
        xg.grad = None
        y.backward(xx, retain_graph=True)
    ms = timeit(ln_bwd)
    print(f"ln bwd (bf16 stream, no residual gradient): {ms:.3f} ms {T * D * 6 / ms / 1e6:.0f} GB/s")
    w = torch.randn(F, D, device=DEV)
    ms = timeit(lambda: ops.cast_weight(w, bf, True, True))
    print(f"cast_weight [{F},{D}]: {ms:.3f} ms")


if __name__ == "__main__":
    main()
