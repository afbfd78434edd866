#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_GZ_ALL=1 MF_GZ_FASTQ=1 timeout -k 5 900 python3 tools/gz_rate.py 8000000 2>&1 | grep -v amdgpu | head -14
