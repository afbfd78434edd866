#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 bench.py > gpurun_out/r05bs_bench_100M.json 2>gpurun_out/r05bs_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05bs_bench_100M.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac']); print(d['end_to_end']); print(d['cli']); print(d['cpu_baseline']['value'], d['slice_restarts'])"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
