#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 600 python3 -m pytest tests/test_round5_gpu.py -q -x -k "cli" > gpurun_out/r05g_cli_tests.txt 2>&1; tail -5 gpurun_out/r05g_cli_tests.txt
bash tools/other_shapes.sh r05g cami_example_k23_b5_l1200 > /dev/null 2>&1
bash tools/other_shapes.sh r05g config5_as_specified | tail -3
MF_IO_TIMING=1 timeout -k 5 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05g_bench_100M.json 2> gpurun_out/r05g_bench.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/r05g_bench_100M.json"))
print(d["ms_per_step"], d["end_to_end"], d["cli"], d["slice_restarts"])
PY
MF_OPTIONS=verbose=2 timeout -k 5 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end 2>&1 >/dev/null | grep -A5 "one-pass what-if" | head -12 > gpurun_out/r05g_one_pass_split_what_if.txt; cat gpurun_out/r05g_one_pass_split_what_if.txt
MF_OPTIONS=verbose=2 timeout -k 5 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --genome-scale 16000000 2>&1 >/dev/null | grep -A5 "one-pass what-if" | head -6 >> gpurun_out/r05g_one_pass_split_what_if.txt
