#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ab() {
  args="$1"; shift
  for o in "$@"; do
    MF_OPTIONS="$o" timeout 300 python3 bench.py $args --no-end-to-end --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('[$args][$o]', d['ms_per_step'], {n:k[n]['ms_per_step'] for n in ('k_skm_count','k_skm_split','k_gather','k_ut_flags','k_cc_adjacency') if n in k})"
  done
}
ab "--reads 50000000 -k 21" "skm_unit_records=2500" "skm_unit_records=3500" "skm_unit_records=4000" "skm_unit_records=5000" "skm_unit_records=4000,skm_unit_distinct=2600"
ab "--reads 50000000 -k 23" "" "skm_unit_records=3000" "skm_unit_records=4000"
ab "--reads 50000000 -k 25" "" "skm_unit_records=3000" "skm_unit_records=4000"
ab "--reads 50000000 -k 27" "" "skm_unit_records=3000"
ab "" "skm_unit_records=3000" "skm_unit_records=1400"
ab "--reads 200000000 -k 21" "" "skm_unit_records=3500"
