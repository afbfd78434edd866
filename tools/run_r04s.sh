set -x
mkdir -p gpurun_out
python -m pytest tests/test_pipeline_gpu.py tests/test_golden_fixture.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r04s_tests.txt 2>&1
tail -4 gpurun_out/r04s_tests.txt
MF_FUZZ_SCALE=25 timeout 300 python3 tools/fuzz.py 90 78 > gpurun_out/r04s_fuzz.txt 2>&1
tail -1 gpurun_out/r04s_fuzz.txt
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04s_bench_100M.json 2> gpurun_out/r04s_bench_100M.err
python3 tools/bench_summary.py gpurun_out/r04s_bench_100M.json | grep "value\|k_cc" | cut -c1-200
MF_OPTIONS=verbose=1 python3 bench.py --samples-per-gpu 8 --reads 200000000 -k 21 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04s_bench_8x200M_k21.json 2> gpurun_out/r04s_bench_8x200M_k21.err
python3 tools/bench_summary.py gpurun_out/r04s_bench_8x200M_k21.json | grep "value\|k_cc\|k_skm" | cut -c1-200
grep "skm:\|arena" gpurun_out/r04s_bench_8x200M_k21.err | tail -40
