set -x
mkdir -p gpurun_out
python -m pytest tests/test_round4_gpu.py -x -q -m gpu > gpurun_out/r04_t4.txt 2>&1
tail -30 gpurun_out/r04_t4.txt
