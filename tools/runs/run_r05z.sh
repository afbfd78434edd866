#!/bin/bash
# final state of round 5: the whole GPU suite, smoke, the driver's bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 3300 python3 -m pytest tests -m gpu -q > gpurun_out/r05z_gpu_tests.txt 2>&1; tail -6 gpurun_out/r05z_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 5 900 python3 bench.py > gpurun_out/r05z_bench_100M.json 2> gpurun_out/r05z_bench.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/r05z_bench_100M.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["end_to_end"]["value"], d["cli"]["value"], d["cli"]["seconds"], d["cpu_baseline"]["value"], d["slice_restarts"])
print(d["roofline_hash_count"]["other_shapes"]["shapes"].keys())
PY
