#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_IO_TIMING=1 timeout -k 5 900 python3 bench.py --reads 20000000 --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/r05bt.err | tail -1 > gpurun_out/r05bt.json
python3 -c "
import json; d=json.load(open('gpurun_out/r05bt.json')); print(d['ms_per_step']); print(d['end_to_end']['seconds']); print(d['cli'])"
grep "count_reads\|driver:" gpurun_out/r05bt.err | tail -4 | cut -c1-260
timeout -k 5 900 python3 -m pytest tests/test_count_gpu.py tests/test_round4_gpu.py -q -x 2>&1 | tail -2
