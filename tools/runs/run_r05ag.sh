#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_VERBOSE=1 timeout -k 5 600 python3 tools/wide_rate.py 50000000 63 2> gpurun_out/r05ag_err.txt > gpurun_out/r05ag_wide_50M_k63.json; cut -c1-900 gpurun_out/r05ag_wide_50M_k63.json; grep count_wide gpurun_out/r05ag_err.txt | head -12
