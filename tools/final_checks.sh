cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r03zz}
timeout -k 5 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/${TAG}_tests.txt
MF_FUZZ_SCALE=25 timeout -k 5 500 python3 tools/fuzz.py 400 51 > gpurun_out/${TAG}_fuzz_large.txt 2>&1
timeout -k 5 300 python3 tools/fuzz.py 200 52 > gpurun_out/${TAG}_fuzz_small.txt 2>&1
timeout -k 5 300 python3 tools/fuzz_cli.py 200 53 > gpurun_out/${TAG}_fuzz_cli.txt 2>&1
bash tools/refresh_profiles.sh ${TAG} > gpurun_out/${TAG}_refresh.log 2>&1
cat gpurun_out/${TAG}_tests.txt; for f in large small cli; do tail -n 1 gpurun_out/${TAG}_fuzz_$f.txt; done; tail -c 300 gpurun_out/${TAG}_bench_100M.json
