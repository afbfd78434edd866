#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_pipeline_gpu.py tests/test_golden_fixture.py tests/test_shapes_gpu.py -q -x 2>&1 | tail -4
timeout -k 5 600 python3 bench.py --no-cpu-baseline --no-end-to-end 2>gpurun_out/r05ay_err.txt | tail -1 > gpurun_out/r05ay_bench.json
python3 -c "
import json; d=json.load(open('gpurun_out/r05ay_bench.json')); print(d['ms_per_step'], d['stage_ms_per_step']); print({k: (round(v['ms_per_step'],1), v['launches']) for k,v in d['kernels'].items() if k.startswith('k_ut')})"
timeout -k 5 600 python3 bench.py --genome-scale 16000000 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05ay_shape_depth_5fold.json
python3 -c "
import json; d=json.load(open('gpurun_out/r05ay_shape_depth_5fold.json')); print(d['ms_per_step'], d['stage_ms_per_step']); print({k: (round(v['ms_per_step'],1), v['launches']) for k,v in d['kernels'].items() if k.startswith('k_ut')})"
