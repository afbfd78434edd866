#!/bin/bash
# experiment: counting units twice as large (one 1024-thread workgroup per CU, 8192-slot LDS table): parity first, then the shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export METAFAST_HIP_LIB=$GRAFT_REPO_ROOT/metafast_amd/lib_big/libmetafast_hip.so
timeout -k 5 900 python3 -m pytest tests/test_count_gpu.py tests/test_round4_gpu.py -q -x --timeout 300 > gpurun_out/r05y_big_tests.txt 2>&1; tail -5 gpurun_out/r05y_big_tests.txt
run() {
  lib=$1; shift
  METAFAST_HIP_LIB=$GRAFT_REPO_ROOT/metafast_amd/$lib/libmetafast_hip.so MF_OPTIONS="$OPT" timeout -k 5 400 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > /tmp/b.json
  python3 - <<PY
import json
try:
    d = json.load(open("/tmp/b.json")); k = d["kernels"]
    print("[$lib][$OPT][$*]", d["ms_per_step"], d["stage_ms_per_step"]["count"], {n: k[n]["ms_per_step"] for n in ("k_skm_count", "k_skm_split", "k_skm_scatter", "k_gather", "k_ut_flags") if n in k}, d["stats"]["n_distinct"], d["stats"]["n_components"])
except Exception as e:
    print("[$lib][$*] failed", e)
PY
}
for lib in lib_big lib; do
  OPT="" run $lib
  OPT="skm_dedupe=0" run $lib
  OPT="" run $lib --genome-scale 16000000
  OPT="" run $lib --reads 50000000 -k 21
done
