#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 2700 python3 -m pytest tests/test_round5_gpu.py -q -x > gpurun_out/r05c_tests.txt 2>&1; tail -15 gpurun_out/r05c_tests.txt
MF_IO_TIMING=1 timeout -k 5 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05c_bench_100M.json 2> gpurun_out/r05c_bench.err
grep "^\[mf\]" gpurun_out/r05c_bench.err | grep -v "arena\|skm pilot\|count(skm)" | tail -40
python3 - <<PY
import json
d = json.load(open("gpurun_out/r05c_bench_100M.json"))
print(d["ms_per_step"], d["end_to_end"], d["cli"])
PY
