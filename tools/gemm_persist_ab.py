#!/usr/bin/env python3
"""Same-process A/B of the persistent ping-pong NT GEMM (vitk_gemm_set_persistent: number of resident
workgroups, 0 = one workgroup per tile) on the eight GEMM instances of a ViT-B/16 training step
(batch 512 by default), interleaved rounds, median of five.
usage: gemm_persist_ab.py [batch] [workgroup counts, comma separated]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,256").split(",")]
T, D, F = B * 197, 768, 3072
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(bf)  # noqa: E731


def case(name, N, K, epi, f32out=False, drop=0.0):
    a, w = r(T, K), r(N, K)
    out = torch.empty(T, N, device=DEV, dtype=torch.float32 if f32out else bf)
    kw = dict(epi=epi)
    if epi in (vitk.EPI_BIAS, vitk.EPI_BIAS_GELU, vitk.EPI_BIAS_RES):
        kw["bias"] = torch.randn(N, device=DEV)
    if epi == vitk.EPI_BIAS_RES:
        kw["residual"] = torch.randn(T, N, device=DEV) if f32out else r(T, N)
    if epi in (vitk.EPI_BIAS_GELU, vitk.EPI_DGELU):
        kw["aux"] = r(T, N)
    if drop:
        kw.update(dropout_p=drop, seed=5)
    return name, 2.0 * T * N * K, (lambda: ops.gemm_nt(a, w, out, **kw))


def main():
    cases = [case("qkv  bias", 3 * D, D, vitk.EPI_BIAS),
             case("proj bias_res_f32", D, D, vitk.EPI_BIAS_RES, True),
             case("fc1  bias_gelu_drop", F, D, vitk.EPI_BIAS_GELU, drop=0.1),
             case("fc2  bias_res_f32_drop", D, F, vitk.EPI_BIAS_RES, True, 0.1),
             case("dfc2 dgelu_drop", F, D, vitk.EPI_DGELU, drop=0.1),
             case("dfc1 none", D, F, vitk.EPI_NONE),
             case("dprj none", D, D, vitk.EPI_NONE),
             case("dqkv none", D, 3 * D, vitk.EPI_NONE)]
    res = {(c[0], s): [] for c in cases for s in STEPS}
    for rnd in range(5):
        for name, fl, fn in cases:
            for s in STEPS:
                vitk.lib.vitk_gemm_set_persistent(s)
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[(name, s)].append(e0.elapsed_time(e1) / 8)
    vitk.lib.vitk_gemm_set_persistent(256)
    print(f"T={T}; median ms (TF/s) per resident-workgroup count (0 = one workgroup per tile)")
    print("instance".ljust(24) + "".join(f"wgs={s}".rjust(16) for s in STEPS))
    for name, fl, _ in cases:
        row = name.ljust(24)
        for s in STEPS:
            ms = statistics.median(res[(name, s)])
            row += f"{ms:8.3f} ({fl / ms / 1e9:5.0f})"
        print(row)


if __name__ == "__main__":
    main()
