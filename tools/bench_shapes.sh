#!/bin/bash
# usage (on the GPU box): tools/bench_shapes.sh <tag>  -> gpurun_out/<tag>_shapes.jsonl : one bench.py line per workload shape
TAG=$1
cd $GRAFT_REPO_ROOT
: > gpurun_out/${TAG}_shapes.jsonl
for a in "--reads 1000000" "--reads 20000000 --read-len 100" "--reads 50000000" "-k 25" "-k 21" "--reads 200000000 -k 21" "--reads 300000000"; do
  timeout -k 5 600 python3 bench.py $a --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${TAG}_shapes.jsonl
done
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_shapes.jsonl"):
    r = json.loads(l); print(r["config"]["workload"][:60], r["ms_per_step"], "%.3e" % r["value"])
PY
