set -x
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r04w_tests.txt 2>&1
tail -25 gpurun_out/r04w_tests.txt
