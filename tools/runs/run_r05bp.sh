#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests/test_files_gpu.py tests/test_round5_gpu.py tests/test_golden_fixture.py -q -x 2>&1 | tail -4
MF_INFLATE_DEBUG=1 timeout -k 5 900 python3 tools/gz_rate.py 20000000 2>&1 | grep -v "^\.bz2\|amdgpu" | tail -12
