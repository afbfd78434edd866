#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_wide_gpu.py -q -x 2>&1 | tail -15
timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>/dev/null > gpurun_out/r05af_wide_200M_k63.json; cut -c1-900 gpurun_out/r05af_wide_200M_k63.json
