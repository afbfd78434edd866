#!/bin/bash
# usage (on the GPU box): tools/probe_shapes.sh <tag>: bench.py lines on parameter sets off the beaten path, looking for kernels that fall over
TAG=$1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for args in "--reads 100000000 -b 5 -l 1200" "--reads 50000000 --sub-rate 0.0005" "--reads 50000000 --sub-rate 0.02" "--reads 50000000 --read-len 100" "--reads 50000000 --read-len 250 -k 25" \
            "--reads 50000000 --b1 5 --b2 50" "--reads 20000000 --genome-scale 100000" "--reads 5000000 --samples-per-gpu 8"; do
  i=$((i+1))
  timeout -k 5 600 python3 bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>gpurun_out/${TAG}_probe_${i}.err | tail -1 > gpurun_out/${TAG}_probe_${i}.json
  python3 - "$args" gpurun_out/${TAG}_probe_${i}.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], "->", d["ms_per_step"], "ms;", d["stage_ms_per_step"])
    print("    ", {k: round(v["ms_per_step"], 1) for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
