cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_pipeline_gpu.py tests/test_golden_fixture.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r03u_tests.txt
timeout -k 5 600 python3 bench.py --no-cpu-baseline > gpurun_out/r03u_bench.json 2> gpurun_out/r03u_bench.err
cat gpurun_out/r03u_tests.txt; python3 -c "
import json
d=json.load(open('gpurun_out/r03u_bench.json')); print(d['ms_per_step'], d['stage_ms_per_step']); print({k:round(v['ms_per_step'],2) for k,v in d['kernels'].items() if 'flags' in k or 'adjac' in k})"
