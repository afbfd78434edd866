#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 300 python3 tools/upload_rate.py 16000000 > gpurun_out/r05d_upload_rate.txt 2>&1; cat gpurun_out/r05d_upload_rate.txt
timeout -k 5 3000 python3 -m pytest tests/test_round5_gpu.py -q -x > gpurun_out/r05d_tests.txt 2>&1; tail -15 gpurun_out/r05d_tests.txt
