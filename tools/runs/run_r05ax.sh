#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_pipeline_gpu.py -q -x -k "unitigs" 2>&1 | tail -12
timeout -k 5 600 python3 bench.py --reads 50000000 -k 23 -b 5 -l 1200 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05ax_shape_cami.json
python3 -c "
import json; d=json.load(open('gpurun_out/r05ax_shape_cami.json')); print(d['ms_per_step'], d['stage_ms_per_step']); print({k: round(v['ms_per_step'],1) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]})"
