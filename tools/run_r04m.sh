set -x
mkdir -p gpurun_out
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04m_bench_100M.json 2> gpurun_out/r04m_bench_100M.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04m_bench_100M.json')); print(d['ms_per_step'], d['end_to_end']); print(d['cli'])"
