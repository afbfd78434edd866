#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python3 bench.py --samples-per-gpu 8 --reads 200000000 -k 21 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05o_bench_8x200M_k21_one_gpu_config4.json
python3 tools/bench_summary.py gpurun_out/r05o_bench_8x200M_k21_one_gpu_config4.json | head -12
python3 -c "
import json; d = json.load(open('gpurun_out/r05o_bench_8x200M_k21_one_gpu_config4.json')); print('slice_restarts', d['slice_restarts'])"
bash tools/other_shapes.sh r05o | tail -8
