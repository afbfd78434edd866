#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 0 1 2 3; do
echo "debug $d (1: large buckets skipped, 2: walk skipped -- wrong tables, timing only)"
MF_WIDE_DEBUG=$d timeout -k 5 600 python3 tools/wide_rate.py 50000000 63 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read())['runs'][1]; print(r['seconds'], r['kernels'])"
done
