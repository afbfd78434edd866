#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_IO_TIMING=1 timeout -k 5 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05ab_bench.json 2> gpurun_out/r05ab_bench.err
grep "^\[mf\]" gpurun_out/r05ab_bench.err | grep "write_kmers\|write_comp\|count_reads\|driver\|ctx_create" | tail -12
python3 -c "
import json; d = json.load(open('gpurun_out/r05ab_bench.json')); print(d['ms_per_step'], d['end_to_end']['seconds'], d['cli'])"
