#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_wide_gpu.py -q -x 2>&1 | tail -3
MF_VERBOSE=0 timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>/dev/null > gpurun_out/r05ap_wide_200M_k63.json; python3 -c "import json,sys; rr=json.load(open('gpurun_out/r05ap_wide_200M_k63.json'))['runs']; print(rr[0]['seconds'], rr[0]['hipmalloc'], rr[1]['seconds'], rr[1]['kernels'])"
