set -x
mkdir -p gpurun_out
python3 tools/cli_rate.py 2 20000000 > gpurun_out/r04h_cli_rate.txt 2>&1
tail -60 gpurun_out/r04h_cli_rate.txt
python -m pytest tests/test_files_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r04h_tests.txt 2>&1
tail -5 gpurun_out/r04h_tests.txt
