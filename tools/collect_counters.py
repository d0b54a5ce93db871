#!/usr/bin/env python3
"""Per-kernel hardware counters of the benchmark step, one rocprofv3 pass per counter group.

    python3 tools/collect_counters.py [--out gpurun_out/r03_counters.json] [--workload vit|swin]

Runs the same short command under rocprofv3 several times (each pass in its own process, behind
its own time limit): a kernel-trace pass for durations, then --pmc passes for FETCH_SIZE,
WRITE_SIZE, the MFMA-busy / busy / wave cycle counters and GRBM_GUI_ACTIVE.  Dispatches are grouped
by kernel (template instance for the GEMMs) and summed; the raw per-dispatch CSVs are deleted, only
the JSON summary is kept.  The summary records a hash of the kernel sources it was measured on;
bench.py reports `roofline.traffic` from it only while that hash still matches the tree.

gfx950 corrections applied here: FETCH_SIZE counts 128-byte requests as 64 bytes, so read bytes =
2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16-byte-per-lane stores and float atomics.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles) with kernel
cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vit-pytorch_amd", "csrc")


def source_hash():
    """sha256 over the kernel sources and the C-ABI header (what the counters depend on)."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))
    files.append(os.path.join(ROOT, "include", "vitk.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _nt(epi, ty):
    return f"gemm_nt_pp<epi{epi},{'f32' if ty in ('float', 'f') else 'bf16'}>"


# rocprofv3 prints some kernel names demangled and some mangled: both spellings are matched
KEYS = [
    (r"gemm_nt_pps?_kernel<\(?(?:int\))?(\d), (float|__bf16)", lambda m: _nt(m.group(1), m.group(2))),
    (r"gemm_nt_pps?_kernelILi(\d)E(DF16b|f)", lambda m: _nt(m.group(1), m.group(2))),
    (r"gemm_nt_bf16_kernelINS_5ShapeILi(\d+)ELi(\d+)E.*?EELi(\d)E", lambda m: f"gemm_nt_{m.group(1)}x{m.group(2)}<epi{m.group(3)}>"),
    (r"gemm_nt_bf16_kernel<.*?Shape<(\d+), (\d+).*?>, (\d)", lambda m: f"gemm_nt_{m.group(1)}x{m.group(2)}<epi{m.group(3)}>"),
    (r"gemm_tn_w4_kernel", lambda m: "gemm_tn_w4"),
    (r"gemm_tn_pp_kernel", lambda m: "gemm_tn_pp"),
    (r"gemm_tn_reduce_kernel", lambda m: "gemm_tn_reduce"),
    (r"gemm_tn_bf16_kernel", lambda m: "gemm_tn_bf16"),
    (r"gemm_\w*f32", lambda m: "gemm_f32"),
    (r"attn_fwd_ws_bf16_kernel", lambda m: "attn_fwd_ws"),
    (r"attn_bwd_ws_bf16_kernel", lambda m: "attn_bwd_ws"),
    (r"win_attn_fwd_mfma_kernel", lambda m: "win_attn_fwd_mfma"),
    (r"win_attn_bwd_mfma_kernel", lambda m: "win_attn_bwd_mfma"),
    (r"(win_attn_\w+?|attn_\w+?)_kernel", lambda m: m.group(1)),
    (r"ln_fwd_kernel", lambda m: "ln_fwd"),
    (r"ln_bwd_reduce_kernel", lambda m: "ln_bwd_reduce"),
    (r"ln_bwd_kernel", lambda m: "ln_bwd"),
    (r"(sgd_\w+?)_kernel", lambda m: m.group(1)),
    (r"(cast_weight|im2col|embed_\w+?|dropout_bwd|convert|colsum|rkr_fold_\w+?|mask_fold_\w+?|scale_\w+?|pb_\w+?|gn_\w+?|wstd_\w+?|add_relu_\w+?|im2col_\w+?|col2im_\w+?)_kernel", lambda m: m.group(1)),
]


def classify(name):
    for pat, fn in KEYS:
        m = re.search(pat, name)
        if m:
            return fn(m)
    return "other (torch / rccl)"


def run_pass(tag, pmc, cmd, outdir, limit):
    d = os.path.join(outdir, tag)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    args = ["timeout", "-s", "KILL", str(limit), "rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p"]
    if pmc:
        args += ["--pmc"] + pmc
    args += ["--"] + cmd
    t0 = time.time()
    print(f"[collect] pass {tag}: {' '.join(pmc) if pmc else 'kernel-trace'}", flush=True)
    r = subprocess.run(args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(f"[collect] pass {tag}: rc={r.returncode} in {time.time() - t0:.0f}s", flush=True)
    if r.returncode != 0:
        print(r.stdout[-2000:], flush=True)
        return None
    return d


def read_trace(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = classify(row["Kernel_Name"])
                e = out.setdefault(k, {"dispatches": 0, "ns": 0.0})
                e["dispatches"] += 1
                e["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return out


def read_counters(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = classify(row["Kernel_Name"])
                e = out.setdefault(k, {})
                c = row["Counter_Name"]
                e[c] = e.get(c, 0.0) + float(row["Counter_Value"])
                e["_n_" + c] = e.get("_n_" + c, 0) + 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_counters.json"))
    ap.add_argument("--workload", default="vit", choices=["vit", "swin"])
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--limit", type=int, default=170, help="seconds per pass")
    args = ap.parse_args()
    if args.workload == "vit":
        cmd = ["python3", "bench.py", "--steps", str(args.steps), "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timer"]
        nsteps = args.steps + 1
    else:
        cmd = ["python3", "tools/swin_bench.py", "imagenet", "--steps", str(args.steps), "--warmup", "1"]
        nsteps = args.steps + 1
    outdir = os.path.join(ROOT, "gpurun_out", "counters_tmp")
    passes = [("trace", None), ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
              ("sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES"]),
              ("grbm", ["GRBM_GUI_ACTIVE"])]
    trace, counters = {}, {}
    for tag, pmc in passes:
        d = run_pass(tag, pmc, cmd, outdir, args.limit)
        if d is None:
            print(f"[collect] pass {tag} failed: stopping here", flush=True)
            break
        if pmc is None:
            trace = read_trace(d)
        else:
            for k, e in read_counters(d).items():
                counters.setdefault(k, {}).update(e)
        shutil.rmtree(d, ignore_errors=True)
    kernels = {}
    for k, t in sorted(trace.items(), key=lambda kv: -kv[1]["ns"]):
        c = counters.get(k, {})
        n = t["dispatches"]
        e = {"dispatches": n, "dispatches_per_step": round(n / nsteps, 2),
             "avg_us": round(t["ns"] / n / 1e3, 2), "ms_per_step": round(t["ns"] / nsteps / 1e6, 3)}
        if "FETCH_SIZE" in c:
            e["read_bytes_per_dispatch"] = int(2 * 1024 * c["FETCH_SIZE"] / c["_n_FETCH_SIZE"])
        if "WRITE_SIZE" in c:
            e["write_bytes_per_dispatch"] = int(1024 * c["WRITE_SIZE"] / c["_n_WRITE_SIZE"])
        if "read_bytes_per_dispatch" in e and "write_bytes_per_dispatch" in e:
            e["hbm_gb_per_s"] = round((e["read_bytes_per_dispatch"] + e["write_bytes_per_dispatch"]) / (t["ns"] / n), 1)
        for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
            if name in c:
                e[name + "_per_dispatch"] = round(c[name] / c["_n_" + name], 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / c["_n_GRBM_GUI_ACTIVE"] / 8.0
            mf = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["_n_SQ_VALU_MFMA_BUSY_CYCLES"]
            e["kernel_cycles"] = round(cyc, 1)
            e["mfma_util"] = round(mf / (4 * 256 * cyc), 4) if cyc > 0 else None
            e["clock_ghz_under_profiler"] = round(cyc / (t["ns"] / n), 3)
        kernels[k] = e
    head = ""
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip()
    except OSError:
        pass
    out = {"command": " ".join(cmd), "steps_in_run": nsteps, "workload": args.workload,
           "source_sha256": source_hash(), "git_head_when_collected": head or None,
           "passes": [t for t, _ in passes],
           "corrections": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-byte requests as 64); write bytes = WRITE_SIZE x 1024; "
                          "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); durations from the kernel-trace pass "
                          "(counter passes serialise dispatches)",
           "kernels": kernels}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    shutil.rmtree(outdir, ignore_errors=True)
    print(f"[collect] wrote {args.out}: {len(kernels)} kernel groups", flush=True)
    for k, e in list(kernels.items())[:14]:
        print(f"  {k:34s} {e['ms_per_step']:8.3f} ms/step  {e.get('hbm_gb_per_s', '-'):>8} GB/s  mfma {e.get('mfma_util', '-')}")


if __name__ == "__main__":
    sys.exit(main())
