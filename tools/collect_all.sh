#!/bin/bash
# Everything the profiles/ directory is refreshed from, on one GPU box (run through gpurun from the repo root):
# counters of the ViT-B/16 and Swin steps, kernel traces of the ViT-B/16, hybrid and Swin steps.  Each step runs
# under its own time limit; the chain stops at the first failure.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
export TMPDIR=/tmp
timeout -k 10 900 python3 tools/collect_counters.py --out gpurun_out/r03_counters.json --workload vit || exit 1
timeout -k 10 900 python3 tools/collect_counters.py --out gpurun_out/r03_swin_counters.json --workload swin || exit 1
cd /tmp || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d "$ROOT/gpurun_out/prof_vit" -o vit -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > "$ROOT/gpurun_out/prof_vit.json" 2> /dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d "$ROOT/gpurun_out/prof_r50" -o r50 -- python3 "$ROOT/bench.py" --model R50_B16 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > "$ROOT/gpurun_out/prof_r50.json" 2> /dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d "$ROOT/gpurun_out/prof_swin" -o swin -- python3 "$ROOT/tools/swin_bench.py" imagenet --steps 4 --warmup 2 > "$ROOT/gpurun_out/prof_swin.txt" 2> /dev/null || exit 1
cd "$ROOT" || exit 1
python3 tools/db_summary.py gpurun_out/prof_vit/vit_results.db 6 "round 3: python bench.py --steps 4 --warmup 2 (ViT-B/16 224 px, batch 512, bf16, N=1 inside the data-parallel wrapper), rocprofv3 --kernel-trace" 40 > gpurun_out/r03_bench_kernel_stats.md || exit 1
python3 tools/db_summary.py gpurun_out/prof_r50/r50_results.db 6 "round 3: python bench.py --model R50_B16 --steps 4 --warmup 2 (R50+ViT-B/16 hybrid, batch 512, bf16), rocprofv3 --kernel-trace" 45 > gpurun_out/r03_r50_kernel_stats.md || exit 1
python3 tools/db_summary.py gpurun_out/prof_swin/swin_results.db 6 "round 3: python tools/swin_bench.py imagenet --steps 4 --warmup 2 (Swin RKR, embed 128, depths [2,2,18], batch 128, bf16), rocprofv3 --kernel-trace" 45 > gpurun_out/r03_swin_kernel_stats.md || exit 1
rm -rf gpurun_out/prof_vit gpurun_out/prof_r50 gpurun_out/prof_swin
echo collected
