set -x
mkdir -p gpurun_out
python -m pytest tests/test_count_gpu.py tests/test_pipeline_gpu.py tests/test_distributed_gpu.py tests/test_golden_fixture.py -x -q -m gpu > gpurun_out/r04d_tests.txt 2>&1
tail -5 gpurun_out/r04d_tests.txt
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --genome-scale 16000000 > gpurun_out/r04d_bench_5x.json 2> gpurun_out/r04d_bench_5x.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04d_bench_100M.json 2> gpurun_out/r04d_bench_100M.err
python3 tools/bench_summary.py gpurun_out/r04d_bench_5x.json | cut -c1-250
python3 tools/bench_summary.py gpurun_out/r04d_bench_100M.json | cut -c1-250
