#!/bin/bash
# builds mf_unitig.o / mf_cc.o with the given -D flags on the GPU box and runs the 100 M-read bench (2 steps)
cd metafast_amd/csrc
for f in mf_unitig mf_cc; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $1 -c $f.hip -o build/$f.o 2>/dev/null; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmetafast_hip.so build/*.o -lpthread -lz -ldl
cd ../..; timeout ${2:-240} python bench.py --no-end-to-end --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], {n:k[n]['ms_per_step'] for n in ('k_ut_flags','k_cc_adjacency')})"
