#!/bin/bash
# usage: tools/gpu_debug.sh <per-stage-timeout> stage...   (log -> gpurun_out/dbg.log)
T=$1; shift
mkdir -p gpurun_out
: > gpurun_out/dbg.log
for s in "$@"; do
  echo "=== stage $s" >> gpurun_out/dbg.log
  timeout -k 5 $T python tools/gpu_debug.py $s >> gpurun_out/dbg.log 2>&1
  echo "=== stage $s rc=$?" >> gpurun_out/dbg.log
done
tail -c 6000 gpurun_out/dbg.log
