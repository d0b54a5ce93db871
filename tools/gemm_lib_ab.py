#!/usr/bin/env python3
"""Same-process A/B of two builds of the kernel library on the GEMM instances of a ViT-B/16 training step
(batch 512 by default): the in-tree libvitk.so against tools/bin/libvitk_exp.so (make -C vit-pytorch_amd/csrc exp
EXP_FLAGS=-D...), interleaved rounds, median of five.
usage: gemm_lib_ab.py [batch] [nt|tn|all] [name=path ...]   (default: exp=tools/bin/libvitk_exp.so)"""
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
WHAT = sys.argv[2] if len(sys.argv) > 2 else "all"
T, D, F = B * 197, 768, 3072
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(bf)  # noqa: E731


def load(path):
    lib = ctypes.CDLL(path)
    for name, args in vitk.PROTOTYPES.items():
        fn = getattr(lib, name, None)          # an older build may lack the newest entry points
        if fn is None:
            continue
        fn.argtypes, fn.restype = args, ctypes.c_int
    lib.vitk_error_string.restype = ctypes.c_char_p
    lib.vitk_error_string.argtypes = [ctypes.c_int]
    return lib


EXTRA = sys.argv[3:] or ["exp=" + os.path.join(ROOT, "tools", "bin", "libvitk_exp.so")]
LIBS = [("base", vitk.lib)] + [(a.split("=", 1)[0], load(a.split("=", 1)[1])) for a in EXTRA]


def nt_case(name, N, K, epi, f32out=False, drop=0.0):
    a, w = r(T, K), r(N, K)
    out = torch.empty(T, N, device="cuda", dtype=torch.float32 if f32out else bf)
    kw = dict(epi=epi)
    if epi in (vitk.EPI_BIAS, vitk.EPI_BIAS_GELU, vitk.EPI_BIAS_RES):
        kw["bias"] = torch.randn(N, device="cuda")
    if epi == vitk.EPI_BIAS_RES:
        kw["residual"] = torch.randn(T, N, device="cuda") if f32out else r(T, N)
    if epi in (vitk.EPI_BIAS_GELU, vitk.EPI_DGELU):
        kw["aux"] = r(T, N)
    if drop:
        kw.update(dropout_p=drop, seed=5)
    return name, 2.0 * T * N * K, (lambda: ops.gemm_nt(a, w, out, **kw))


def tn_case(name, M, N):
    dy, x = r(T, M), r(T, N)
    c = torch.zeros(M, N, device="cuda")
    db = torch.zeros(M, device="cuda")
    return name, 2.0 * T * M * N, (lambda: ops.gemm_tn_acc(dy, x, c, 0, colsum=db))


def main():
    cases = []
    if WHAT in ("nt", "all"):
        cases += [nt_case("qkv  bias", 3 * D, D, vitk.EPI_BIAS),
                  nt_case("proj bias_res_f32", D, D, vitk.EPI_BIAS_RES, True),
                  nt_case("fc1  bias_gelu_drop", F, D, vitk.EPI_BIAS_GELU, drop=0.1),
                  nt_case("fc2  bias_res_f32_drop", D, F, vitk.EPI_BIAS_RES, True, 0.1),
                  nt_case("dfc2 dgelu_drop", F, D, vitk.EPI_DGELU, drop=0.1),
                  nt_case("dfc1 none", D, F, vitk.EPI_NONE),
                  nt_case("dprj none", D, D, vitk.EPI_NONE),
                  nt_case("dqkv none", D, 3 * D, vitk.EPI_NONE)]
    if WHAT in ("tn", "all"):
        cases += [tn_case("dW qkv", 3 * D, D), tn_case("dW proj", D, D), tn_case("dW fc1", F, D), tn_case("dW fc2", D, F)]
    for _, lib in LIBS:          # the weight-gradient workspace is registered per library
        ops.lib = lib
        ops._TN_WORKSPACE.clear()
        for _, _, fn in cases:
            fn()
    # (the second library keeps the workspace tensor of the first alive through this list)
    res = {}
    for rnd in range(5):
        for name, fl, fn in cases:
            for lname, lib in (LIBS if rnd % 2 == 0 else LIBS[::-1]):      # the later of a pair runs hotter: alternate the order
                ops.lib = lib
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((name, lname), []).append(e0.elapsed_time(e1) / 8)
    ops.lib = vitk.lib
    print(f"T={T}; median ms (TF/s)")
    tot = {l: 0.0 for l, _ in LIBS}
    for name, fl, _ in cases:
        row = name.ljust(24)
        for lname, _ in LIBS:
            ms = statistics.median(res[(name, lname)])
            tot[lname] += ms
            row += f"  {lname} {ms:7.3f} ({fl / ms / 1e9:5.0f})"
        print(row)
    print("sum".ljust(24) + "".join(f"  {l} {tot[l]:7.3f}        " for l, _ in LIBS))


if __name__ == "__main__":
    main()
