set -x
mkdir -p gpurun_out
python -m pytest tests/test_pipeline_gpu.py tests/test_golden_fixture.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r04r_tests.txt 2>&1
tail -4 gpurun_out/r04r_tests.txt
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04r_bench_100M.json 2> gpurun_out/r04r_bench_100M.err
python3 tools/bench_summary.py gpurun_out/r04r_bench_100M.json | grep "value\|k_cc" | cut -c1-200
MF_OPTIONS=verbose=1 python3 bench.py --samples-per-gpu 8 --reads 200000000 -k 21 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04r_bench_8x200M_k21.json 2> gpurun_out/r04r_bench_8x200M_k21.err
python3 tools/bench_summary.py gpurun_out/r04r_bench_8x200M_k21.json | grep "value\|k_cc\|k_skm" | cut -c1-200
grep "skm:\|count(skm): n_occ\|pilot:\|components: thr\|arena" gpurun_out/r04r_bench_8x200M_k21.err | tail -70
