#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 2400 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r05av_gpu_tests.txt; cat gpurun_out/r05av_gpu_tests.txt
timeout -k 5 900 python3 bench.py > gpurun_out/r05av_bench_100M.json 2>gpurun_out/r05av_bench_err.txt; cut -c1-400 gpurun_out/r05av_bench_100M.json
