set -x
mkdir -p gpurun_out
python -m pytest tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r04f_tests.txt 2>&1
tail -15 gpurun_out/r04f_tests.txt
MF_IO_TIMING=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04f_bench_100M.json 2> gpurun_out/r04f_bench_100M.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04f_bench_100M.json')); print(d['ms_per_step'], d['end_to_end']); print(d['cli'])"
grep "\[mf\]" gpurun_out/r04f_bench_100M.err | tail -40
