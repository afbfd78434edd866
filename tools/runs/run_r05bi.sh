#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_TRY_RCCL_2RANKS=1 timeout -k 5 600 python3 -m pytest tests/test_distributed_gpu.py -q -x -k "rccl or RCCL or two_ranks" 2>&1 | tail -15
