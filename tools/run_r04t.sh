set -x
mkdir -p gpurun_out
MF_OPTIONS=verbose=1 python3 bench.py --samples-per-gpu 8 --reads 200000000 -k 21 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04t_bench_8x200M_k21.json 2> gpurun_out/r04t_bench_8x200M_k21.err
python3 tools/bench_summary.py gpurun_out/r04t_bench_8x200M_k21.json | grep "value\|k_cc\|k_skm" | cut -c1-200
grep "skm:\|arena: hipMalloc" gpurun_out/r04t_bench_8x200M_k21.err | tail -30
