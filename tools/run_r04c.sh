set -x
mkdir -p gpurun_out
python -m pytest tests/test_count_gpu.py -x -q -m gpu -k "depth_independent" > gpurun_out/r04c_tests.txt 2>&1
tail -5 gpurun_out/r04c_tests.txt
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --genome-scale 16000000 > gpurun_out/r04c_bench_5x.json 2> gpurun_out/r04c_bench_5x.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04c_bench_100M.json 2> gpurun_out/r04c_bench_100M.err
python3 tools/bench_summary.py gpurun_out/r04c_bench_5x.json gpurun_out/r04c_bench_100M.json
