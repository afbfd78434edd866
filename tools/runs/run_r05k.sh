#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for o in "ablate=32" "ablate=32,skm_dedupe=0"; do
  echo "== $o"
  MF_OPTIONS=$o,verbose=1 timeout -k 5 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end 2>&1 >/dev/null | grep "wave cycles\|count(skm): n_occ" | head -30
done
echo "== 5-fold depth"
MF_OPTIONS=ablate=32,verbose=1 timeout -k 5 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --genome-scale 16000000 2>&1 >/dev/null | grep "wave cycles\|count(skm): n_occ" | head -30
