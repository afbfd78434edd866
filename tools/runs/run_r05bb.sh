#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "--reads 50000000 --read-len 250 -k 25" "--reads 50000000 --read-len 250 -k 31" "--reads 30000000 --read-len 400 -k 21"; do
  MF_OPTIONS=verbose=1 timeout -k 5 600 python3 bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>gpurun_out/r05bb.err | tail -1 > gpurun_out/r05bb.json
  python3 - "$args" gpurun_out/r05bb.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "->", d["ms_per_step"], "ms;", d["stage_ms_per_step"])
print("    ", {k: round(v["ms_per_step"], 1) for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]})
PY
  grep "count(skm): n_occ\|skm pilot: 0" gpurun_out/r05bb.err | head -3 | cut -c1-200
done
timeout -k 5 1200 python3 -m pytest tests/test_pipeline_gpu.py tests/test_count_gpu.py tests/test_round4_gpu.py -q -x 2>&1 | tail -3
