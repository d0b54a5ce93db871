#!/bin/bash
# usage: tools/gpurun_wait.sh <outfile> <timeout> '<command>'
# Runs gpurun; while the pool has no free box (exit status 3: nothing ran, nothing charged) waits and asks again.
out=$1; to=$2; shift 2
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$out" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
