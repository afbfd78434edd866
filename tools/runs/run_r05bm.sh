#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 tools/gz_rate.py 5000000 2>&1 | tail -4
timeout -k 5 900 python3 -m pytest tests/test_files_gpu.py -q -x 2>&1 | tail -3
