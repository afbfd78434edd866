#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_INFLATE_DEBUG=1 timeout -k 5 900 python3 tools/gz_rate.py 20000000 2>&1 | grep "inflate\|^\.gz\|^two" | tail -8
