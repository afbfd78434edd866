#!/bin/bash
# usage (on the GPU box): tools/pmc_wide.sh <tag> [reads]  -> gpurun_out/<tag>_traffic_k63.json: the PMC passes of tools/refresh_profiles.sh on the
# k = 63 extension (one sample, reads resident): which of VALU, LDS and HBM the record path's kernels wait for
TAG=$1; READS=${2:-50000000}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for spec in "c|FETCH_SIZE" "d|WRITE_SIZE" "a|GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "b|GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  L=${spec%%|*}; CNT=${spec#*|}
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d gpurun_out/pmc_${TAG}w_$L -o p -- python3 bench.py -k 63 --reads $READS --steps 1 --warmup 0 > gpurun_out/pmc_${TAG}w_$L.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}w gpurun_out/${TAG}_traffic_k63.json > /dev/null
python3 - <<PY
import json
t = json.load(open("gpurun_out/${TAG}_traffic_k63.json"))
for k, v in sorted(t.items(), key=lambda kv: -kv[1]["hbm_GB"])[:14]:
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "note"})
PY
