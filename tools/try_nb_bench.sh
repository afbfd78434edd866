#!/bin/bash
# tools/try_nb_bench.sh "<-D flags>" [timeout s] [bench args]: builds the library with the given flags THROUGH the Makefile (resource
# guard included) into /tmp/mf_try -- never into the shipped library -- and runs the 100 M-read bench (3 steps) on it.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out=/tmp/mf_try_$(echo "$1" | md5sum | cut -c1-8)
make -C "$HERE/metafast_amd/csrc" -j8 BUILD=$out/build OUTDIR=$out EXTRA="$1" > $out.log 2>&1 || { tail -5 $out.log; echo "[$1] build refused"; exit 1; }
cd "$HERE"; METAFAST_HIP_LIB=$out/libmetafast_hip.so timeout ${2:-240} python3 bench.py --no-end-to-end --no-cpu-baseline --steps 3 --warmup 1 $3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('[$1]', d['ms_per_step'], {n:k[n]['ms_per_step'] for n in ('k_ut_flags','k_cc_adjacency','k_skm_count','k_skm_scatter','k_skm_split') if n in k})"
