#!/bin/bash
# second batch of tools/probe_shapes.sh
TAG=$1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for args in "--reads 100000000 --read-len 75" "--reads 50000000 -k 20" "--reads 50000000 -k 19" "--reads 50000000 -k 15" "--reads 50000000 -b 20" \
            "--reads 50000000 --genome-scale 100000000" "--reads 2000000 --samples-per-gpu 16" "--reads 50000000 -k 30" "--reads 50000000 -k 24 -l 500"; do
  i=$((i+1))
  timeout -k 5 600 python3 bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>gpurun_out/${TAG}_probe_${i}.err | tail -1 > gpurun_out/${TAG}_probe_${i}.json
  python3 - "$args" gpurun_out/${TAG}_probe_${i}.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], "->", d["ms_per_step"], "ms;", d["value"] / 1e9, "G k-mers/s;", d["stage_ms_per_step"])
    print("    ", {k: round(v["ms_per_step"], 1) for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
