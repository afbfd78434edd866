#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_wide_gpu.py -q -x -k "config4 or 20M" --durations=3 2>&1 | tail -12
