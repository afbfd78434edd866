#!/bin/bash
# usage (on the GPU box): tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>_* : bench line, rocprofv3 kernel stats of the same command, PMC traffic passes
TAG=$1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 400 python3 bench.py > gpurun_out/${TAG}_bench_100M.json 2> gpurun_out/${TAG}_bench.err
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o p -- python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_rocprof.err
timeout -k 5 300 tools/pmc_bench.sh ${TAG} 100000000 c FETCH_SIZE
timeout -k 5 300 tools/pmc_bench.sh ${TAG} 100000000 d WRITE_SIZE
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG} gpurun_out/${TAG}_traffic_100M.json > /dev/null
ls gpurun_out/${TAG}_stats/ | head; tail -c 400 gpurun_out/${TAG}_bench_100M.json
