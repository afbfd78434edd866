#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests/test_distributed_gpu.py tests/test_pipeline_gpu.py tests/test_bench_gpu.py -q -x > gpurun_out/r05w_tests.txt 2>&1; tail -4 gpurun_out/r05w_tests.txt
MF_FORCE_DIST=1 timeout -k 5 300 python3 bench.py --reads 50000000 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05w_bench_50M_rccl_world1.json
python3 -c "
import json; d = json.load(open('gpurun_out/r05w_bench_50M_rccl_world1.json')); print(d['ms_per_step'], d['comm'], d['stage_ms_per_step'])"
