set -x
mkdir -p gpurun_out
MF_IO_TIMING=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04n_bench_20M.json 2> gpurun_out/r04n_bench_20M.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04n_bench_20M.json')); print(d['ms_per_step'], d['end_to_end']); print(d['cli'])"
grep "\[mf\]" gpurun_out/r04n_bench_20M.err | tail -32
python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -3
