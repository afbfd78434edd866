cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_count_gpu.py -x -q -m gpu 2>&1 | tail -3
MF_VERBOSE=0 timeout -k 5 600 python3 tools/count_ab.py 100000000 31 3072:skm_dedupe=1,3072:skm_dedupe=0 2>&1 | grep "part_target\|signature" | tail -8
timeout -k 5 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03s_bench.json 2> gpurun_out/r03s_bench.err
python3 - <<'PY'
import json
for f in ("gpurun_out/r03s_bench.json",):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["stage_ms_per_step"]["count"]); k=d["kernels"]
    for n in ("k_skm_scatter","k_skm_split","k_skm_count"): print(n, k[n]["launches"], k[n]["ms_per_step"])
PY
