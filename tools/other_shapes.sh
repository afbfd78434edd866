#!/bin/bash
# usage (on the GPU box): tools/other_shapes.sh <tag>  -> gpurun_out/<tag>_shape_*.json (one bench.py line per workload) and
# gpurun_out/<tag>_hash_count_other_shapes.json (what bench.py carries as roofline_hash_count.other_shapes once it is copied to
# profiles/hash_count_other_shapes.json): the hash-count kernel's priced roofline fraction on the shapes that are not the headline one
TAG=$1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ONLY=${2:-}
for spec in "dedupe_off|MF_OPTIONS=skm_dedupe=0|" "k21_50M||--reads 50000000 -k 21" "k23_50M_cami_k||--reads 50000000 -k 23" "depth_5fold||--genome-scale 16000000" \
            "cami_example_k23_b5_l1200||--reads 50000000 -k 23 -b 5 -l 1200" "config5_as_specified|MF_OVERLAP_SAMPLES=0|--samples-per-gpu 4 --reads 120000000 --pool-scale 5700000 --sub-rate 0.01" \
            "config5_as_specified_overlapped|MF_OVERLAP_SAMPLES=1|--samples-per-gpu 4 --reads 120000000 --pool-scale 5700000 --sub-rate 0.01"; do
  # (several samples per GPU: the default runs them one after the other; MF_OVERLAP_SAMPLES=1 -- sample i's unitigs beside sample i + 1's count on a
  # second context -- is the experiment of round 6: kernels of two streams that run side by side stretch each other's event times, the step gains nothing)
  name=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; args=${rest#*|}
  if [ -n "$ONLY" ] && [ "$ONLY" != "$name" ]; then continue; fi
  env $envs timeout -k 5 900 python3 bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/${TAG}_shape_${name}.json
done
python3 - <<PY
import json, glob, os
out = {"source": "bench.py lines run by tools/other_shapes.sh ${TAG} on one MI355X (builder-run; profiles/${TAG}_shape_*.json)", "shapes": {}}
for p in sorted(glob.glob("gpurun_out/${TAG}_shape_*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(p, "unreadable", e); continue
    r = d["roofline_hash_count"]
    name = os.path.basename(p)[len("${TAG}_shape_"):-5]
    out["shapes"][name] = dict(workload=d["config"]["workload"], genome_scale_bp=d["config"]["genome_scale_bp"], substitutions_per_base=d["config"]["substitutions_per_base"],
                               options={"dedupe_off": "skm_dedupe=0", "config5_as_specified_overlapped": "MF_OVERLAP_SAMPLES=1"}.get(name, ""),
                               frac=r["frac"], launch_ms_per_step=r["launch_ms"], priced_GB=r["algorithmic_GB"], ms_per_step=d["ms_per_step"], value=d["value"])
    print(name, r["frac"], r["launch_ms"], d["ms_per_step"])
json.dump(out, open("gpurun_out/${TAG}_hash_count_other_shapes.json", "w"), indent=1)
PY
