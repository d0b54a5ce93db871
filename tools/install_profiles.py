#!/usr/bin/env python3
"""Copy what tools/collect_all.sh left in gpurun_out/ into profiles/ (run here, after the GPU call): the counter files
get the current commit hash when their source hash still matches the tree, the NT-instance table is rebuilt."""
import glob
import hashlib
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "vit-pytorch_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    files.append(os.path.join(ROOT, "include", "vitk.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def nt_table(d):
    M = 512 * 197
    gf = lambda n, k, m=M: 2.0 * m * n * k / 1e9  # noqa: E731
    inst = {
        "gemm_nt_pp<epi1,bf16>": ("QKV projection, bias (N=2304, K=768)", [gf(2304, 768)], 623),
        "gemm_nt_pp<epi3,f32>": ("out-proj / fc2, bias + fp32 residual in, fp32 out (N=768; K=768 and 3072)", [gf(768, 768), gf(768, 3072)], (776 + 1245) / 2),
        "gemm_nt_pp<epi2,bf16>": ("fc1, bias + GELU + dropout, writes g and the GELU derivative (N=3072, K=768)", [gf(3072, 768)], 1400),
        "gemm_nt_pp<epi4,bf16>": ("dX of fc2 times the saved GELU derivative / dropout factor (N=3072, K=768)", [gf(3072, 768)], 1395),
        "gemm_nt_pp<epi0,bf16>": ("dX of fc1, out-proj, QKV (N=768; K=3072, 768, 2304)", [gf(768, 3072), gf(768, 768), gf(768, 2304)], (775 + 310 + 620) / 3),
        "gemm_nt_pp<epi5,f32>": ("patch embedding + position embedding + dropout (M=100352, N=768, K=768)", [gf(768, 768, 512 * 196)], 0),
    }
    out = ["# round 3: the NT GEMM instances inside the ViT-B/16 step (batch 512, bf16), from profiles/r03_counters.json\n",
           "Durations from the kernel-trace pass, bytes from the FETCH_SIZE / WRITE_SIZE passes (gfx950-corrected), MFMA utilisation =",
           "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles).  `alg MB` = operands + results + epilogue operands of one call.\n",
           "| instance | what | calls/step | avg us | GF/call | TF/s | of 2.5 PF/s | read MB | write MB | alg MB | MFMA util |",
           "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    tot_f = tot_t = 0.0
    for k, (what, fl, alg) in inst.items():
        e = d["kernels"].get(k)
        if not e:
            continue
        f = sum(fl) / len(fl)
        tf = f / (e["avg_us"] * 1e-6) / 1e3
        tot_f += f * e["dispatches_per_step"]
        tot_t += e["ms_per_step"]
        out.append("| `%s` | %s | %.0f | %.1f | %.0f | %.0f | %.3f | %.0f | %.0f | %s | %.3f |" % (
            k, what, e["dispatches_per_step"], e["avg_us"], f, tf, tf / 2500, e["read_bytes_per_dispatch"] / 1e6,
            e["write_bytes_per_dispatch"] / 1e6, ("%.0f" % alg) if alg else "-", e["mfma_util"]))
    out.append("\nAll NT instances of the ping-pong kernel together: %.1f TFLOP in %.2f ms per step = %.0f TF/s (%.3f of peak) on the "
               "device this was collected on." % (tot_f / 1e3, tot_t, tot_f / tot_t, tot_f / tot_t / 2500))
    tn_name = "gemm_tn_w4" if "gemm_tn_w4" in d["kernels"] else "gemm_tn_pp"
    e = d["kernels"].get(tn_name)
    if e:
        fw = 12 * (gf(2304, 768) + gf(768, 768) + gf(3072, 768) + gf(768, 3072)) + gf(768, 768, 512 * 196)
        red = d["kernels"].get("gemm_tn_reduce", {"ms_per_step": 0.0})
        t = e["ms_per_step"] + red["ms_per_step"]
        out.append("Weight-gradient GEMM (`%s` + `gemm_tn_reduce`): %.1f TFLOP in %.2f ms per step = %.0f TF/s (%.3f of peak), "
                   "MFMA utilisation of the product kernel %.3f, %.0f MB read + %.0f MB written per call." % (
                       tn_name, fw / 1e3, t, fw / t, fw / t / 2500, e["mfma_util"], e["read_bytes_per_dispatch"] / 1e6,
                       e["write_bytes_per_dispatch"] / 1e6))
    return "\n".join(out) + "\n"


def main():
    head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip()
    h = source_hash()
    for name in ("r03_counters.json", "r03_swin_counters.json"):
        src = os.path.join(ROOT, "gpurun_out", name)
        with open(src) as fh:
            d = json.load(fh)
        ok = d["source_sha256"] == h
        d["git_head_when_collected"] = head if ok else None
        with open(os.path.join(ROOT, "profiles", name), "w") as fh:
            json.dump(d, fh, indent=1)
        print(name, "source hash matches the tree:", ok)
        if name == "r03_counters.json":
            with open(os.path.join(ROOT, "profiles", "r03_nt_instances.md"), "w") as fh:
                fh.write(nt_table(d))
    for f in ("r03_bench_kernel_stats.md", "r03_r50_kernel_stats.md", "r03_swin_kernel_stats.md"):
        shutil.copy(os.path.join(ROOT, "gpurun_out", f), os.path.join(ROOT, "profiles", f))


if __name__ == "__main__":
    main()
