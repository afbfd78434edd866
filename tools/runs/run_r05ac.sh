#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 3300 python3 -m pytest tests -m gpu -q > gpurun_out/r05ac_gpu_tests.txt 2>&1; tail -5 gpurun_out/r05ac_gpu_tests.txt
timeout -k 5 300 python3 tools/fuzz_cli.py 200 21 > gpurun_out/r05ac_fuzz_cli.txt 2>&1; tail -1 gpurun_out/r05ac_fuzz_cli.txt
timeout -k 5 300 python3 tools/fuzz.py 200 22 > gpurun_out/r05ac_fuzz.txt 2>&1; tail -1 gpurun_out/r05ac_fuzz.txt
