#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1200 python3 -m pytest tests/test_files_gpu.py tests/test_round4_gpu.py tests/test_round5_gpu.py -q -x -k "not other_k and not cami" > gpurun_out/r05i_tests.txt 2>&1; tail -5 gpurun_out/r05i_tests.txt
MF_IO_TIMING=1 timeout -k 5 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05i_bench_100M.json 2> gpurun_out/r05i_bench.err
grep "^\[mf\]" gpurun_out/r05i_bench.err | grep -v "arena\|skm pilot\|count(skm)\|skm:" | sed -n 6,40p
python3 - <<PY
import json
d = json.load(open("gpurun_out/r05i_bench_100M.json"))
print(d["ms_per_step"], d["end_to_end"], d["cli"], d["slice_restarts"])
PY
