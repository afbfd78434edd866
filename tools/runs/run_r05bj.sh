#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_OPTIONS=ut_double_after=1 timeout -k 5 400 python3 tools/fuzz.py 240 501 2>&1 | tail -3
timeout -k 5 300 python3 tools/fuzz.py 150 502 2>&1 | tail -2
timeout -k 5 300 python3 tools/fuzz_files.py 120 503 2>&1 | tail -2
