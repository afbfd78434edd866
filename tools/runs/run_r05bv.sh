#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 400 python3 tools/fuzz_files.py 240 611 2>&1 | tail -2
timeout -k 5 2700 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r05bv_gpu_tests.txt; cat gpurun_out/r05bv_gpu_tests.txt
