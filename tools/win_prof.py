#!/usr/bin/env python3
"""Where the window-attention backward spends its clocks: per-phase cycle counts of the two waves of a workgroup,
averaged over workgroups, for the three Swin stage geometries at batch 128.
Build the instrumented library first: make -C vit-pytorch_amd/csrc abl   (-> tools/bin/libvitk_abl.so)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VITK_LIB"] = os.path.join(ROOT, "tools", "bin", "libvitk_abl.so")
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402

raw = ctypes.CDLL(os.environ["VITK_LIB"])
PH = ["prologue", "staging", "scores+products+stores", "-", "table flush"]


def timed(qkv, table, B, H, heads, shift, go, reps=5):
    outs = [ops.window_attention(qkv, table, B, H, H, heads, 7, shift) for _ in range(reps)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for o in outs:
        o.backward(go)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def ablations(B):
    """Backward time with parts of the kernel left out (VITK_WIN_ABL bits; results are wrong, only the time is read)."""
    names = {0: "complete", 1: "no LDS atomics", 2: "no bias-table reads", 4: "no global stores", 8: "no transposed images",
             16: "no exp", 31: "none of these"}
    for (H, C, heads) in ((56, 128, 4), (14, 512, 16)):
        qkv = torch.randn(B * H * H, 3 * C, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        table = torch.randn(169, heads, device="cuda", requires_grad=True)
        go = torch.randn(B * H * H, C, device="cuda", dtype=torch.bfloat16)
        for groups in (1024, 2048, 4096):
            os.environ["VITK_WIN_GROUPS"] = str(groups)
            row = []
            for bits, nm in names.items():
                os.environ["VITK_WIN_ABL"] = str(bits)
                timed(qkv, table, B, H, heads, 3, go, 2)
                row.append(f"{nm} {timed(qkv, table, B, H, heads, 3, go):.0f}")
            print(f"{H}x{H} heads {heads}, {groups} workgroups: backward us: " + ", ".join(row))
    os.environ.pop("VITK_WIN_ABL")
    os.environ.pop("VITK_WIN_GROUPS")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    ablations(B)
    for (H, C, heads) in ((56, 128, 4), (28, 256, 8), (14, 512, 16)):
        qkv = torch.randn(B * H * H, 3 * C, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        table = torch.randn(169, heads, device="cuda", requires_grad=True)
        for shift in (0, 3):
            out = ops.window_attention(qkv, table, B, H, H, heads, 7, shift)
            go = torch.randn_like(out)
            out.backward(go)                                  # warm-up
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 16)()
            raw.vitk_debug_win_prof(buf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            out = ops.window_attention(qkv, table, B, H, H, heads, 7, shift)
            torch.cuda.synchronize()
            e0.record()
            out.backward(go)
            e1.record()
            torch.cuda.synchronize()
            raw.vitk_debug_win_prof(buf)
            nwin = B * (H // 7) ** 2
            groups = min(nwin, max(4096 // heads, 1))
            nwg = groups * heads
            print(f"{H}x{H} heads {heads} shift {shift}: backward (all kernels) {e0.elapsed_time(e1) * 1e3:.0f} us, {nwg} workgroups, "
                  f"{buf[5] / nwg:.1f} windows each")
            for w in range(2):
                row = ", ".join(f"{PH[k]} {buf[8 * w + k] / nwg:.0f}" for k in range(5))
                print(f"   wave {w} clocks per workgroup: {row}; total {sum(buf[8 * w + k] for k in range(5)) / nwg:.0f}")


if __name__ == "__main__":
    main()
