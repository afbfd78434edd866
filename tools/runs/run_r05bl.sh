#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 tools/gz_rate.py 5000000 2>&1 | tail -5
