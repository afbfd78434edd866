cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_golden_fixture.py tests/test_files_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r03o_tests.txt
timeout -k 5 600 python3 bench.py --no-cpu-baseline > gpurun_out/r03o_bench.json 2> gpurun_out/r03o_bench.err
MF_OPTIONS=ut_split_links=1 timeout -k 5 600 python3 bench.py --no-cpu-baseline > gpurun_out/r03o_bench_split.json 2> gpurun_out/r03o_bench_split.err
cat gpurun_out/r03o_tests.txt; python3 -c "
import json
for f in ('gpurun_out/r03o_bench.json','gpurun_out/r03o_bench_split.json'):
    d=json.load(open(f)); print(d['ms_per_step'], d['stage_ms_per_step']); print({k:round(v['ms_per_step'],2) for k,v in d['kernels'].items() if k.startswith('k_ut') or 'scatter' in k})"
