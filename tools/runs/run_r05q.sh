#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() {
  lib=$1; shift
  echo "== $lib $*"
  METAFAST_HIP_LIB=$GRAFT_REPO_ROOT/metafast_amd/$lib/libmetafast_hip.so timeout -k 5 400 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > /tmp/b.json
  python3 - <<PY
import json
try:
    d = json.load(open("/tmp/b.json"))
    k = d["kernels"]
    print(d["ms_per_step"], d["stage_ms_per_step"]["count"], d["stats"]["n_records"], d["stats"]["n_distinct"], d["stats"]["n_components"], {n: k[n]["ms_per_step"] for n in ("k_skm_scatter", "k_skm_split", "k_skm_count", "k_ut_flags", "k_cc_adjacency") if n in k})
except Exception as e:
    print("failed", e)
PY
}
run lib_m14 --reads 50000000 -k 25
run lib --reads 50000000 -k 25
run lib_m14 --reads 50000000 -k 27
run lib --reads 50000000 -k 27
run lib_m14 --reads 50000000 -k 29
run lib --reads 50000000 -k 29
run lib_m14 --reads 50000000 -k 30
run lib --reads 50000000 -k 30
