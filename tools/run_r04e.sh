set -x
mkdir -p gpurun_out
python -m pytest tests/test_count_gpu.py -x -q -m gpu > gpurun_out/r04e_tests.txt 2>&1
tail -3 gpurun_out/r04e_tests.txt
MF_OPTIONS=skm_dynq=0 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04e_bench_100M_static.json 2> gpurun_out/r04e_bench_100M_static.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04e_bench_100M.json 2> gpurun_out/r04e_bench_100M.err
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --genome-scale 16000000 > gpurun_out/r04e_bench_5x.json 2> gpurun_out/r04e_bench_5x.err
for f in r04e_bench_100M_static r04e_bench_100M r04e_bench_5x; do python3 tools/bench_summary.py gpurun_out/$f.json | grep "value\|k_skm_count " | cut -c1-250; done
