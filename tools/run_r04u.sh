set -x
mkdir -p gpurun_out
python -m pytest tests/test_wide_gpu.py -x -q -m gpu > gpurun_out/r04u_tests.txt 2>&1
tail -6 gpurun_out/r04u_tests.txt
