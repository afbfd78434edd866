#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 3300 python3 -m pytest tests -m gpu -q -x > gpurun_out/r05n_gpu_tests.txt 2>&1; tail -8 gpurun_out/r05n_gpu_tests.txt
for a in "--reads 50000000 -k 21" "--reads 50000000 -k 23" "--reads 50000000 -k 25" ""; do
  timeout -k 5 400 python3 bench.py $a --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05n_bench_$(echo $a | tr -d ' -').json
  python3 tools/bench_summary.py gpurun_out/r05n_bench_$(echo $a | tr -d ' -').json | head -1
done
