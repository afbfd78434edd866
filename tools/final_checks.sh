cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r03w_tests.txt
MF_FUZZ_SCALE=25 timeout -k 5 500 python3 tools/fuzz.py 400 11 > gpurun_out/r03w_fuzz_large.txt 2>&1
timeout -k 5 300 python3 tools/fuzz.py 200 12 > gpurun_out/r03w_fuzz_small.txt 2>&1
timeout -k 5 300 python3 tools/fuzz_cli.py 200 13 > gpurun_out/r03w_fuzz_cli.txt 2>&1
bash tools/refresh_profiles.sh r03y > gpurun_out/r03y_refresh.log 2>&1
cat gpurun_out/r03w_tests.txt; for f in large small cli; do tail -n 1 gpurun_out/r03w_fuzz_$f.txt; done; tail -c 300 gpurun_out/r03y_bench_100M.json
