set -x
mkdir -p gpurun_out
python -m pytest tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r04a_dist_tests.txt 2>&1
tail -3 gpurun_out/r04a_dist_tests.txt
python3 tools/plan_sweep.py 100000000 16000000 auto 10,11 11,11 10,10 > gpurun_out/r04a_sweep_5x.txt 2>&1
python3 tools/plan_sweep.py 100000000 4000000 auto 10,11 11,11 > gpurun_out/r04a_sweep_21x.txt 2>&1
python3 tools/plan_sweep.py 100000000 1000000 auto 10,11 9,10 > gpurun_out/r04a_sweep_83x.txt 2>&1
python3 tools/plan_sweep.py 50000000 random auto 11,11 10,11 > gpurun_out/r04a_sweep_random.txt 2>&1
cat gpurun_out/r04a_sweep_*.txt
