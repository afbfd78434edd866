cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_count_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r03n_tests.txt
timeout -k 5 600 python3 tools/count_ab.py 100000000 31 6144 > gpurun_out/r03n_count_ab.txt 2>&1
timeout -k 5 600 python3 tools/count_ab.py 50000000 21 6144 > gpurun_out/r03n_count_ab_k21.txt 2>&1
cat gpurun_out/r03n_tests.txt; grep -v "^\[mf\] count\|amdgpu.ids\|several passes" gpurun_out/r03n_count_ab.txt gpurun_out/r03n_count_ab_k21.txt
