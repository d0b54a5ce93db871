#!/usr/bin/env python3
"""In-kernel clock and phase cycles of the ping-pong NT GEMM (diagnostic build: make -C
vit-pytorch_amd/csrc abl).  For each shape the kernel runs back to back on random data for about
two seconds, then the stamps of the last launch are read: per workgroup the shader-clock cycles of
prologue / main loop / epilogue and the clock the chip held (shader cycles per 100 MHz tick).
The main loop's floor is 2048 cycles per 64-deep K-tile (128 MFMAs of 16 cycles per SIMD).
Usage: python tools/gemm_stamp.py [B]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VITK_LIB"] = os.path.join(ROOT, "tools", "bin", "libvitk_abl.so")
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    M = (B * 197) // 256 // 256 * 256 * 256 or 256 * 256    # whole rounds of 256 CUs only: one ping-pong launch
    raw = ctypes.CDLL(os.environ["VITK_LIB"])
    raw.vitk_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    bf = torch.bfloat16
    vitk.lib.vitk_debug_force_gemm_shape(4)
    raw.vitk_debug_gemm_abl.argtypes = [ctypes.c_int]
    cases = [("fc2", 768, 3072, "none", 0), ("fc1", 3072, 768, "none", 0), ("fc1", 3072, 768, "gelu", 0),
             ("proj", 768, 768, "none", 0), ("proj", 768, 768, "res", 0)]
    # loop ablations (results are garbage, cycles are not): 1 no LDS-DMA issue, 2 no fragment reads, 4 no barriers (per-wave
    # stamps are then not comparable), 8 no waits for landed data;
    # 16 / 32 / 64: nt / sc0 / sc1 cache policy on the LDS-DMA loads (valid results)
    if os.environ.get("GEMM_STAMP_ABL", "1") != "0" and os.environ.get("GEMM_STAMP_EPI", "0") == "0":
        cases += [("fc2", 768, 3072, "none", m) for m in (1, 2, 3, 8, 16, 32, 64, 96)]
        cases += [("fc1", 3072, 768, "none", m) for m in (16, 32, 64, 96)]
    # epilogue ablations (mask << 8): 1 no arithmetic, 2 no global stores, 4 no loads of the extra operand, 8 no LDS turn-around
    # and no stores (results garbage)
    if os.environ.get("GEMM_STAMP_EPI", "0") != "0":
        cases = []
        for case in (("fc1", 3072, 768, "gelu"), ("dfc2", 3072, 768, "dgelu"), ("proj", 768, 768, "res32"), ("fc2", 768, 3072, "res32")):
            cases += [case + (m << 8,) for m in (0, 1, 2, 3, 4, 6, 8, 9)]
    for name, N, K, epi, abl in cases:
        raw.vitk_debug_gemm_abl(abl)
        a = torch.randn(M, K, device="cuda").to(bf)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
        out = torch.empty(M, N, device="cuda", dtype=bf)
        bias = torch.randn(N, device="cuda")
        aux = torch.empty(M, N, device="cuda", dtype=bf)
        res = torch.randn(M, N, device="cuda").to(bf)
        if epi == "none":
            fn = lambda: ops.gemm_nt(a, w, out)
        elif epi == "gelu":
            fn = lambda: ops.gemm_nt(a, w, out, epi=vitk.EPI_BIAS_GELU, bias=bias, aux=aux, dropout_p=0.1, seed=5)
        elif epi == "dgelu":
            fn = lambda: ops.gemm_nt(a, w, out, epi=vitk.EPI_DGELU, aux=aux, dropout_p=0.1, seed=5)
        elif epi == "res32":
            out32 = torch.empty(M, N, device="cuda")
            res32 = torch.randn(M, N, device="cuda")
            fn = lambda: ops.gemm_nt(a, w, out32, epi=vitk.EPI_BIAS_RES, bias=bias, residual=res32, dropout_p=0.1, seed=5)
        else:
            fn = lambda: ops.gemm_nt(a, w, out, epi=vitk.EPI_BIAS_RES, bias=bias, residual=res)
        t_end = time.time() + 2.0
        n = 0
        while time.time() < t_end:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n += 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nwg = (M // 256) * ((N + 255) // 256)
        nst = min(nwg, 4096)
        st = np.zeros((nst, 10), dtype=np.uint64)
        rc = raw.vitk_debug_gemm_stamps(st.ctypes.data, nst)
        assert rc == 0, rc
        c = st[:, :4].astype(np.int64)
        r = st[:, 4:8].astype(np.int64)
        # turnover: per CU (XCC id + SE/SH/CU bits of HW_ID), time between the end of one workgroup and the
        # entry of the next one that ran there, in 10 ns ticks of the 100 MHz counter
        cu = ((st[:, 9].astype(np.int64) & 0xF) << 8) | ((st[:, 8].astype(np.int64) >> 8) & 0xFF)
        gaps, per_cu = [], []
        for key in np.unique(cu):
            idx = np.where(cu == key)[0]
            idx = idx[np.argsort(r[idx, 0])]
            per_cu.append(len(idx))
            gaps += list(r[idx[1:], 0] - r[idx[:-1], 3])
        gaps = np.array(gaps) if gaps else np.zeros(1)
        span = (r[:, 3].max() - r[:, 0].min()) * 0.01
        turn = (f"turnover per CU: {len(per_cu)} CUs x {np.median(per_cu):.0f} WGs, gap median {np.median(gaps) * 0.01:.2f} us "
                f"p90 {np.percentile(gaps, 90) * 0.01:.2f} us; WG life median {np.median(r[:, 3] - r[:, 0]) * 0.01:.2f} us; "
                f"first entry -> last exit {span:.1f} us")
        total_c, total_r = c[:, 3] - c[:, 0], r[:, 3] - r[:, 0]
        ok = total_r > 0
        clk = np.median(total_c[ok] / total_r[ok]) * 0.1          # GHz
        loop_clk = np.median((c[:, 2] - c[:, 1])[ok] / np.maximum(r[:, 2] - r[:, 1], 1)[ok]) * 0.1
        pro, loop, ep = (np.median(c[:, 1] - c[:, 0]), np.median(c[:, 2] - c[:, 1]), np.median(c[:, 3] - c[:, 2]))
        floor = (K // 64) * 2048
        tf = 2.0 * M * N * K / ms / 1e9
        print(f"{name:4s} N={N:4d} K={K:4d} {epi:4s} abl={abl}: {ms:.3f} ms {tf:5.0f} TF/s | clock {clk:.2f} GHz (loop {loop_clk:.2f}) | "
              f"cycles/WG prologue {pro:.0f} loop {loop:.0f} ({floor / loop * 100:.0f} % of MFMA floor {floor}) epilogue {ep:.0f} | "
              f"peak at this clock {256 * 4 * 1024 * clk / 1e3:.0f} TF/s | {turn}", flush=True)
    raw.vitk_debug_gemm_abl(0)
    vitk.lib.vitk_debug_force_gemm_shape(-1)


if __name__ == "__main__":
    main()
