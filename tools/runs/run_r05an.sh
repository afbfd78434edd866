#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_VERBOSE=1 timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>gpurun_out/r05an_err.txt >/dev/null; grep "count_wide" gpurun_out/r05an_err.txt | head -5
