set -x
mkdir -p gpurun_out
python -m pytest tests/test_pipeline_gpu.py tests/test_golden_fixture.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r04x_tests.txt 2>&1
tail -3 gpurun_out/r04x_tests.txt
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r04x_bench_100M.json 2> gpurun_out/r04x_bench_100M.err
python3 tools/bench_summary.py gpurun_out/r04x_bench_100M.json | grep "value\|k_ut_flags\|k_cc_adj" | cut -c1-200
