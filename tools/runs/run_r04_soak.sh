set -x
mkdir -p gpurun_out
MF_FUZZ_SCALE=25 timeout 400 python3 tools/fuzz.py 300 501 > gpurun_out/r04b_fuzz_large.txt 2>&1; tail -1 gpurun_out/r04b_fuzz_large.txt
timeout 400 python3 tools/fuzz.py 300 502 > gpurun_out/r04b_fuzz_small.txt 2>&1; tail -1 gpurun_out/r04b_fuzz_small.txt
timeout 300 python3 tools/fuzz_cli.py 240 503 > gpurun_out/r04b_fuzz_cli.txt 2>&1; tail -1 gpurun_out/r04b_fuzz_cli.txt
timeout 200 python3 tools/fuzz_files.py 150 504 > gpurun_out/r04b_fuzz_files.txt 2>&1; tail -1 gpurun_out/r04b_fuzz_files.txt
python3 bench.py --samples-per-gpu 4 --reads 380000000 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04b_bench_4x380M.json 2> /dev/null
python3 tools/bench_summary.py gpurun_out/r04b_bench_4x380M.json | head -1
python3 bench.py --samples-per-gpu 4 --reads 120000000 --pool-scale 5700000 --sub-rate 0.01 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04b_bench_4x120M_c5.json 2> /dev/null
python3 tools/bench_summary.py gpurun_out/r04b_bench_4x120M_c5.json | head -1
python3 bench.py --reads 50000000 -k 21 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04b_bench_50M_k21.json 2> /dev/null
python3 tools/bench_summary.py gpurun_out/r04b_bench_50M_k21.json | head -1
