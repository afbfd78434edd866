#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() {
  lib=$1; opt=$2; shift 2
  METAFAST_HIP_LIB=$GRAFT_REPO_ROOT/metafast_amd/$lib/libmetafast_hip.so MF_OPTIONS="$opt,verbose=1" timeout -k 5 400 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/tmp/err.txt | tail -1 > /tmp/b.json
  python3 - <<PY
import json, re
d = json.load(open("/tmp/b.json")); k = d["kernels"]
redo = sum(int(x) for x in re.findall(r"batch \d+: (\d+) partition\(s\) counted in several passes", open("/tmp/err.txt").read()))
print("[$lib][$opt][$*]", d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("k_skm_count", "k_skm_split", "k_gather", "k_ut_flags") if n in k}, "units redone (all steps):", redo)
PY
}
for lib in lib lib_p2; do
  for o in "skm_unit_distinct=2200" "skm_unit_distinct=2600" "skm_unit_distinct=3000" "skm_unit_distinct=3400"; do run $lib $o; done
  run $lib "skm_unit_distinct=2200" --genome-scale 16000000
  run $lib "skm_unit_distinct=3000" --genome-scale 16000000
done
