#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_wide_gpu.py -q -x 2>&1 | tail -3
MF_WIDE_DEBUG=0 timeout -k 5 600 python3 tools/wide_rate.py 100000000 63 2>/dev/null | python3 -c "import json,sys; rr=json.loads(sys.stdin.read())['runs']; print(rr[1]['seconds'], rr[1]['kernels'])"
