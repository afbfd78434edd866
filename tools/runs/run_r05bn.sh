#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 tools/gz_rate.py 5000000 2>&1 | grep -v "^\.bz2\|amdgpu" | tail -4
echo "zlib only (MF_FAST_INFLATE=0):"
MF_FAST_INFLATE=0 timeout -k 5 900 python3 tools/gz_rate.py 5000000 2>&1 | grep "^\.gz\|two" | tail -2
timeout -k 5 900 python3 -m pytest tests/test_files_gpu.py tests/test_inflate_cpu.py -q -x 2>&1 | tail -3
