#!/bin/bash
# timing ablations of the neighbour lookup (results are WRONG with NB_ABLATE != 0): rebuilds mf_unitig.o on the GPU box
cd metafast_amd/csrc
for a in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNB_ABLATE=$a -c mf_unitig.hip -o build/mf_unitig.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmetafast_hip.so build/*.o -lpthread -lz -ldl
  (cd ../..; python bench.py --no-end-to-end --steps 2 --warmup 1 2>/tmp/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('ablate $a', d['ms_per_step'], {n:k[n]['ms_per_step'] for n in k if n.startswith('k_ut_')})")
  tail -3 /tmp/err.txt
done
