#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "-k 23" "--read-len 250"; do
MF_IO_TIMING=1 timeout -k 5 900 python3 bench.py --reads 20000000 $args --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/r05bg.err | tail -1 > gpurun_out/r05bg.json
echo "== $args"; grep "^\[mf\]" gpurun_out/r05bg.err | grep -v "arena" | tail -25 | cut -c1-260
done
