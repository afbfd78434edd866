#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_round5_gpu.py -q -x -k "long_reads" 2>&1 | tail -8
