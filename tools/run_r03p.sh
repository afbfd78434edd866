cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_count_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r03p_tests.txt
timeout -k 5 600 python3 tools/count_ab.py 100000000 31 6144,6144:ablate=32 > gpurun_out/r03p_count_ab.txt 2>&1
timeout -k 5 600 python3 tools/count_ab.py 50000000 21 6144 > gpurun_out/r03p_count_ab_k21.txt 2>&1
MF_FUZZ_SCALE=1 timeout -k 5 600 python3 tools/fuzz.py 400 > gpurun_out/r03p_fuzz.txt 2>&1
cat gpurun_out/r03p_tests.txt; grep -v "^\[mf\] count(skm): slice\|amdgpu.ids\|several passes" gpurun_out/r03p_count_ab.txt gpurun_out/r03p_count_ab_k21.txt; tail -3 gpurun_out/r03p_fuzz.txt
