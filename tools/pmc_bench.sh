#!/bin/bash
# usage: tools/pmc_bench.sh <tag> <reads> <pass-letter> "<counters>"  -> gpurun_out/pmc_<tag>_<letter>/ (one rocprofv3 --pmc pass over bench.py)
TAG=$1; READS=${2:-20000000}; L=$3; CNT=$4
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d gpurun_out/pmc_${TAG}_$L -o p -- python3 bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > gpurun_out/pmc_${TAG}_$L.log 2>&1
