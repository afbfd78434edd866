#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_wide_gpu.py -q -x 2>&1 | tail -3
for d in 0 1; do
echo "ablate $d (1: large buckets skipped, 2: representatives' walk skipped -- wrong tables, timing only)"
MF_WIDE_DEBUG=$d timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>/dev/null | python3 -c "import json,sys; rr=json.loads(sys.stdin.read())['runs']; print(rr[0]['seconds'], rr[0]['hipmalloc'], rr[1]['seconds'], rr[1]['kernels'])"
done
