#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_wide_gpu.py -q -x 2>&1 | tail -8
for d in 0 2; do
echo "debug $d (2: walk skipped -- wrong tables, timing only)"
MF_WIDE_DEBUG=$d timeout -k 5 600 python3 tools/wide_rate.py 50000000 63 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read())['runs'][1]; print(r['seconds'], r['kernels'])"
done
timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>/dev/null > gpurun_out/r05ai_wide_200M_k63.json; cut -c1-900 gpurun_out/r05ai_wide_200M_k63.json
