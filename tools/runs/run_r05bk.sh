#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1200 python3 -m pytest tests/test_count_gpu.py tests/test_round4_gpu.py tests/test_golden_fixture.py -q -x 2>&1 | tail -3
for o in "skm_digests=1" "skm_digests=0"; do
MF_OPTIONS=$o timeout -k 5 600 python3 bench.py --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05bk.json
python3 - "$o" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05bk.json"))
print(sys.argv[1], "->", d["ms_per_step"], "ms;", d["stage_ms_per_step"]["count"], {k: round(v["ms_per_step"], 2) for k, v in d["kernels"].items() if k.startswith("k_skm")})
PY
done
