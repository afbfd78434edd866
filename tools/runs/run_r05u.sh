#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests/test_round4_gpu.py tests/test_count_gpu.py tests/test_round5_gpu.py -q -x -k "not cli and not device_parser" > gpurun_out/r05u_tests.txt 2>&1; tail -4 gpurun_out/r05u_tests.txt
ab() {
  args="$1"; shift
  for o in "$@"; do
    MF_OPTIONS="$o" timeout 300 python3 bench.py $args --no-end-to-end --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('[$args][$o]', d['ms_per_step'], {n:k[n]['ms_per_step'] for n in ('k_skm_count','k_skm_split','k_gather') if n in k})"
  done
}
ab "--reads 100000000 -k 21" "" "skm_unit_records=2000"
ab "--reads 20000000 -k 21" "" "skm_unit_records=2000"
ab "--reads 50000000 -k 22" "" "skm_unit_records=2000"
ab "--reads 50000000 -k 24" "" "skm_unit_records=2000"
