cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_wide_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r03s_tests.txt
cat gpurun_out/r03s_tests.txt
timeout -k 5 120 python3 __graft_entry__.py smoke 2>&1 | tail -2
bash tools/refresh_profiles.sh r03z 2>&1 | tail -3
