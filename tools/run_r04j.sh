set -x
mkdir -p gpurun_out
python3 tools/cli_rate.py 2 20000000 > gpurun_out/r04j_cli_rate.txt 2>&1
grep "total\|\[mf\]" gpurun_out/r04j_cli_rate.txt | tail -45
python -m pytest tests/test_files_gpu.py tests/test_pipeline_gpu.py tests/test_golden_fixture.py -x -q -m gpu > gpurun_out/r04j_tests.txt 2>&1
tail -5 gpurun_out/r04j_tests.txt
python3 tools/fuzz_cli.py 120 41 > gpurun_out/r04j_fuzz_cli.txt 2>&1
tail -3 gpurun_out/r04j_fuzz_cli.txt
