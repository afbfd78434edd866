#!/bin/bash
# usage (on the GPU box): tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>_* : bench line, rocprofv3 kernel stats of the same command,
# PMC passes (HBM traffic: FETCH_SIZE, WRITE_SIZE; SQ: VALU / LDS activity) -> gpurun_out/<tag>_traffic_100M.json (what bench.py reads as
# profiles/traffic_100M.json: roofline.traffic and roofline.bound come from it, nothing is edited by hand), then the bench line again with it
TAG=$1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o p -- python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_rocprof.err
timeout -k 5 300 tools/pmc_bench.sh ${TAG} 100000000 c FETCH_SIZE
timeout -k 5 300 tools/pmc_bench.sh ${TAG} 100000000 d WRITE_SIZE
timeout -k 5 300 tools/pmc_bench.sh ${TAG} 100000000 a "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"
timeout -k 5 300 tools/pmc_bench.sh ${TAG} 100000000 b "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG} gpurun_out/${TAG}_traffic_100M.json > /dev/null
cp gpurun_out/${TAG}_traffic_100M.json profiles/traffic_100M.json
timeout -k 5 500 python3 bench.py > gpurun_out/${TAG}_bench_100M.json 2> gpurun_out/${TAG}_bench.err
ls gpurun_out/${TAG}_stats/ | head; tail -c 600 gpurun_out/${TAG}_bench_100M.json
