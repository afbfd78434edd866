#!/bin/bash
# timing ablations of the neighbour lookup (results are WRONG with NB_ABLATE != 0).  Every variant is built THROUGH the Makefile
# (so check_resources.py's no-scratch guard runs on it: a spilling k_ut_flags_part hung the GPU in round 4) into a build
# directory and a library of its own under /tmp; the shipped metafast_amd/lib/libmetafast_hip.so and csrc/build are never touched.
# bench.py loads the variant through METAFAST_HIP_LIB; every run is under a timeout.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for a in 1 2 3; do
  out=/tmp/mf_ab_nb$a
  make -C "$HERE/metafast_amd/csrc" -j8 BUILD=$out/build OUTDIR=$out EXTRA="-DNB_ABLATE=$a" > $out.log 2>&1 || { tail -5 $out.log; echo "ablate $a: build refused"; continue; }
  (cd "$HERE"; METAFAST_HIP_LIB=$out/libmetafast_hip.so timeout 300 python3 bench.py --no-end-to-end --no-cpu-baseline --steps 2 --warmup 1 2>/tmp/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('ablate $a', d['ms_per_step'], {n:k[n]['ms_per_step'] for n in k if n.startswith('k_ut_')})")
  tail -3 /tmp/err.txt
done
