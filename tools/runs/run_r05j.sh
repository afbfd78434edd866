#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 300 python3 tools/upload_rate.py 20000000 > gpurun_out/r05j_upload_rate.txt 2>&1; cat gpurun_out/r05j_upload_rate.txt
