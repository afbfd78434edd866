#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_VERBOSE=1 timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>gpurun_out/r05aq_err.txt > gpurun_out/r05aq_wide_200M_k63.json; python3 -c "import json,sys; rr=json.load(open('gpurun_out/r05aq_wide_200M_k63.json'))['runs']; print(rr[0]['seconds'], rr[0]['hipmalloc'], rr[1]['seconds'], rr[1]['kernels'])"; grep count_wide gpurun_out/r05aq_err.txt | head -5
