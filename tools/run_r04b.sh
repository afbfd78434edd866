set -x
mkdir -p gpurun_out
export VERBOSE=1
python3 tools/plan_sweep.py 100000000 16000000 auto > gpurun_out/r04b_sweep_5x.txt 2>&1
python3 tools/plan_sweep.py 100000000 4000000 auto > gpurun_out/r04b_sweep_21x.txt 2>&1
python3 tools/plan_sweep.py 100000000 1000000 auto > gpurun_out/r04b_sweep_83x.txt 2>&1
python3 tools/plan_sweep.py 50000000 random auto > gpurun_out/r04b_sweep_random.txt 2>&1
python3 tools/plan_sweep.py 1000000 random auto > gpurun_out/r04b_sweep_random1M.txt 2>&1
grep -h "pilot\|plan\|count(skm): n_occ" gpurun_out/r04b_sweep_*.txt
python -m pytest tests/test_count_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r04b_tests.txt 2>&1
tail -3 gpurun_out/r04b_tests.txt
