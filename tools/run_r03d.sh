cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_count_gpu.py tests/test_pipeline_gpu.py tests/test_golden_fixture.py tests/test_distributed_gpu.py tests/test_files_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r03d_tests.txt
timeout -k 5 600 python3 tools/count_ab.py 100000000 31 6144,3072,8192 > gpurun_out/r03d_count_ab.txt 2>&1
timeout -k 5 600 python3 bench.py --no-cpu-baseline > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err
cat gpurun_out/r03d_tests.txt; grep -v "^\[mf\] count\|amdgpu.ids" gpurun_out/r03d_count_ab.txt; cat gpurun_out/r03d_bench.json
