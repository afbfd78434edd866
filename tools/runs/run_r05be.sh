#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_pipeline_gpu.py -q -x -k "unitigs" 2>&1 | tail -5
for args in "--reads 50000000 -b 20" "--reads 50000000 -k 23 -b 5 -l 1200" "--reads 100000000 -b 5 -l 1200"; do
  timeout -k 5 600 python3 bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05be.json
  python3 - "$args" gpurun_out/r05be.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "->", d["ms_per_step"], "ms;", d["stage_ms_per_step"])
print("    ", {k: round(v["ms_per_step"], 1) for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]) if k.startswith("k_ut")})
PY
done
