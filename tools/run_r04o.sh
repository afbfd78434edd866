set -x
mkdir -p gpurun_out
python3 bench.py --samples-per-gpu 4 --reads 120000000 --pool-scale 5700000 --sub-rate 0.01 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04o_bench_config5_as_specified.json 2> gpurun_out/r04o_bench_config5.err
python3 tools/bench_summary.py gpurun_out/r04o_bench_config5_as_specified.json | cut -c1-300
python3 -c "
import json; d=json.load(open('gpurun_out/r04o_bench_config5_as_specified.json')); print(d['stats'])"
tail -5 gpurun_out/r04o_bench_config5.err
