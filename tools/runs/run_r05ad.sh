#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 600 python3 tools/sort_rate.py 100000000 > gpurun_out/r05ad_sort_rate.txt 2>&1; cat gpurun_out/r05ad_sort_rate.txt
MF_IO_TIMING=1 timeout -k 5 300 python3 tools/cli_rate.py 2 20000000 2>&1 | grep "write_kmers\|write_components" | tail -4
