#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "-k 23 -b 5 -l 1200" "-b 20" "--read-len 250"; do
timeout -k 5 900 python3 bench.py --reads 20000000 $args --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/r05bf.err | tail -1 > gpurun_out/r05bf.json
python3 - "$args" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05bf.json"))
print(sys.argv[1], "->", d["ms_per_step"], "ms;"); print("   e2e", d["end_to_end"]); print("   cli", d["cli"])
PY
done
