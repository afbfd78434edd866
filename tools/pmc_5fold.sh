#!/bin/bash
# usage (on the GPU box): tools/pmc_5fold.sh <tag>  -> gpurun_out/<tag>_traffic_5fold.json: the PMC passes of tools/refresh_profiles.sh on the 5-fold-depth
# shape (100 M reads of a 16 x larger pool: --genome-scale 16000000), VERDICT r5 item 7
TAG=$1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for spec in "c|FETCH_SIZE" "d|WRITE_SIZE" "a|GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "b|GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  L=${spec%%|*}; CNT=${spec#*|}
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d gpurun_out/pmc_${TAG}5_$L -o p -- python3 bench.py --genome-scale 16000000 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > gpurun_out/pmc_${TAG}5_$L.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}5 gpurun_out/${TAG}_traffic_5fold.json > /dev/null
python3 - <<PY
import json
t = json.load(open("gpurun_out/${TAG}_traffic_5fold.json"))
for k, v in sorted(t.items(), key=lambda kv: -kv[1]["hbm_GB"])[:16]:
    print(k, v["hbm_GB"], v.get("valu_busy"), v.get("lds_busy"))
PY
