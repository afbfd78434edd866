#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 tools/gz_rate.py 40000000 2>&1 | grep "^\.gz\|^two\|^metafast\|inflated\|driver" | tail -8
