cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_count_gpu.py tests/test_files_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r03b_tests.txt
timeout -k 5 600 python3 tools/count_ab.py 100000000 31 6144,6144:count_variant=1,6144:ablate=32,6144:ablate=32:count_variant=1 > gpurun_out/r03b_count_ab.txt 2>&1
timeout -k 5 600 python -m pytest tests/test_config5_gpu.py -x -q -m gpu --durations=3 2>&1 | tail -12 > gpurun_out/r03b_config5.txt
cat gpurun_out/r03b_tests.txt gpurun_out/r03b_config5.txt; grep -v "^\[mf\] count\|amdgpu.ids" gpurun_out/r03b_count_ab.txt
