#!/bin/bash
# measurements at HEAD after the per-k minimizer length and the per-k unit bound: shapes, config 4, a short fuzz soak
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python3 bench.py --samples-per-gpu 8 --reads 200000000 -k 21 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05v_bench_8x200M_k21_one_gpu_config4.json
python3 tools/bench_summary.py gpurun_out/r05v_bench_8x200M_k21_one_gpu_config4.json | head -1
python3 -c "
import json; d = json.load(open('gpurun_out/r05v_bench_8x200M_k21_one_gpu_config4.json')); print('slice_restarts', d['slice_restarts'])"
bash tools/other_shapes.sh r05v | tail -8
timeout -k 5 400 python3 tools/fuzz.py 240 11 > gpurun_out/r05v_fuzz.txt 2>&1; tail -2 gpurun_out/r05v_fuzz.txt
timeout -k 5 200 python3 tools/fuzz_files.py 90 12 > gpurun_out/r05v_fuzz_files.txt 2>&1; tail -1 gpurun_out/r05v_fuzz_files.txt
timeout -k 5 300 python3 tools/fuzz_cli.py 150 13 > gpurun_out/r05v_fuzz_cli.txt 2>&1; tail -1 gpurun_out/r05v_fuzz_cli.txt
