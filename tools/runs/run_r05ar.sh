#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 4 12 28; do
echo "ablate $d (k_wide_big: 4 = no write-back, 8 = no ranking, 16 = no hash inserts -- wrong tables, timing only)"
MF_WIDE_DEBUG=$d timeout -k 5 600 python3 tools/wide_rate.py 100000000 63 2>/dev/null | python3 -c "import json,sys; rr=json.loads(sys.stdin.read())['runs']; print(rr[1]['seconds'], rr[1]['kernels'])"
done
MF_WIDE_DEBUG=0 timeout -k 5 600 python3 tools/wide_rate.py 100000000 63 2>/dev/null | python3 -c "import json,sys; rr=json.loads(sys.stdin.read())['runs']; print(rr[1]['seconds'], rr[1]['kernels'])"
