#!/bin/bash
# usage: tools/prof_step.sh <tag> [bench args]: kernel trace of a short bench run condensed to gpurun_out/<tag>_kernels.md
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
tag=$1; shift
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d "$ROOT/gpurun_out/prof_$tag" -o $tag -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer "$@" > "$ROOT/gpurun_out/prof_$tag.json" 2> /dev/null || exit 1
cd "$ROOT" || exit 1
python3 tools/db_summary.py gpurun_out/prof_$tag/${tag}_results.db 6 "$tag: python bench.py --steps 4 --warmup 2 $*, rocprofv3 --kernel-trace" 24 > gpurun_out/${tag}_kernels.md || exit 1
rm -rf gpurun_out/prof_$tag
