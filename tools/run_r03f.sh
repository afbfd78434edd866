cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_golden_fixture.py tests/test_distributed_gpu.py tests/test_files_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r03f_tests.txt
timeout -k 5 600 python3 bench.py --no-cpu-baseline > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
timeout -k 5 1800 python -m pytest tests/test_shapes_gpu.py -x -q -m gpu --durations=6 2>&1 | tail -14 > gpurun_out/r03f_shapes.txt
cat gpurun_out/r03f_tests.txt gpurun_out/r03f_shapes.txt; python3 -c "
import json; d=json.load(open('gpurun_out/r03f_bench.json')); print(d['ms_per_step'], d['stage_ms_per_step']); print({k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
