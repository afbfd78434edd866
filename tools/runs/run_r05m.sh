#!/bin/bash
# experiment: minimizer length 13 instead of 15 (variant library in metafast_amd/lib_m13) on the k = 21 / k = 23 shapes, with a parity check against the shipped library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in metafast_amd/lib/libmetafast_hip.so metafast_amd/lib_m13/libmetafast_hip.so; do
  for a in "--reads 50000000 -k 21" "--reads 50000000 -k 23"; do
    echo "== $lib $a"
    METAFAST_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 5 400 python3 bench.py $a --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > /tmp/b.json
    python3 - <<PY
import json
d = json.load(open("/tmp/b.json"))
k = d["kernels"]
print(d["ms_per_step"], d["stage_ms_per_step"], d["stats"]["n_records"], d["stats"]["n_distinct"], d["stats"]["n_components"], {n: k[n]["ms_per_step"] for n in ("k_skm_scatter", "k_skm_split", "k_skm_count", "k_ut_flags", "k_cc_adjacency", "k_gather") if n in k})
print("matrix/vec checksum:", d["stats"])
PY
  done
done
