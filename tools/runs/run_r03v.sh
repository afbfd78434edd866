cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_count_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 5 150 python3 tools/fuzz.py 100 71 2>&1 | tail -n 1
for dd in 1 0; do
MF_OPTIONS=skm_dedupe=$dd timeout -k 5 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03v_bench.json 2> gpurun_out/r03v_bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03v_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], "count", d["kernels"]["k_skm_count"]["ms_per_step"])
PY
done
