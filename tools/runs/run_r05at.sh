#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_VERBOSE=1 timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>gpurun_out/r05at_err.txt > gpurun_out/r05at_wide_200M_k63.json; python3 -c "import json,sys; rr=json.load(open('gpurun_out/r05at_wide_200M_k63.json'))['runs']; print(rr[0]['seconds'], rr[0]['hipmalloc'], rr[1]['seconds'], rr[1]['kernels'])"; grep count_wide gpurun_out/r05at_err.txt | head -5 > gpurun_out/r05at_wide_200M_k63_passes.txt
timeout -k 5 600 python3 tools/wide_rate.py 200000000 47 2>/dev/null > gpurun_out/r05at_wide_200M_k47.json; python3 -c "import json,sys; rr=json.load(open('gpurun_out/r05at_wide_200M_k47.json'))['runs']; print(rr[1]['seconds'], rr[1]['kmers_per_s'], rr[1]['kernels'])"
timeout -k 5 600 python3 tools/wide_rate.py 200000000 33 2>/dev/null > gpurun_out/r05at_wide_200M_k33.json; python3 -c "import json,sys; rr=json.load(open('gpurun_out/r05at_wide_200M_k33.json'))['runs']; print(rr[1]['seconds'], rr[1]['kmers_per_s'], rr[1]['kernels'])"
