#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 600 python3 tools/nbr_locality.py 4000000 300000 > gpurun_out/r05h_nbr_locality.txt 2>&1; cat gpurun_out/r05h_nbr_locality.txt | tail -6
MF_IO_TIMING=1 timeout -k 5 300 python3 tools/cli_rate.py 2 20000000 2>&1 | grep "load_components\|features\|exit\|total" | tail
