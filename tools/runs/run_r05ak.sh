#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>/dev/null > gpurun_out/r05ak_wide_200M_k63.json; cut -c1-1200 gpurun_out/r05ak_wide_200M_k63.json
