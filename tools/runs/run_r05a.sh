#!/bin/bash
# round 5, first GPU call: the new tests, then the shapes the k-specialised neighbour kernels are for
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests/test_round5_gpu.py -x -q > gpurun_out/r05a_tests.txt 2>&1; tail -15 gpurun_out/r05a_tests.txt
for a in "--reads 50000000 -k 21" "--reads 50000000 -k 23" ""; do
  timeout -k 5 400 python3 bench.py $a --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05a_bench_$(echo $a | tr -d ' -').json
  python3 tools/bench_summary.py gpurun_out/r05a_bench_$(echo $a | tr -d ' -').json
done
