#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_OPTIONS=verbose=1 timeout -k 5 600 python3 bench.py --reads 50000000 --read-len 250 -k 25 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end 2>gpurun_out/r05ba.err | tail -1 > gpurun_out/r05ba.json
grep "^\[mf\]" gpurun_out/r05ba.err | grep -v arena | head -30
