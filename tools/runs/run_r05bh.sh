#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 2700 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r05bh_gpu_tests.txt; cat gpurun_out/r05bh_gpu_tests.txt
