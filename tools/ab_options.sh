#!/bin/bash
# A/B of context options on the 100 M-read bench: tools/ab_options.sh "<name=value,...>" ...   (one bench run per argument; "" = defaults)
for o in "$@"; do
  MF_OPTIONS="$o" timeout 300 python3 bench.py --no-end-to-end --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('[$o]', d['ms_per_step'], {n:k[n]['ms_per_step'] for n in ('k_ut_flags','k_cc_adjacency','k_skm_count','k_gather','k_index_build_part','k_ut_contract')})"
done
