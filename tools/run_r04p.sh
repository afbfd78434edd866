set -x
mkdir -p gpurun_out
MF_OPTIONS=verbose=1 python3 bench.py --samples-per-gpu 8 --reads 50000000 -k 21 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > gpurun_out/r04p_bench_8x50M_k21.json 2> gpurun_out/r04p_bench_8x50M_k21.err
grep "components: thr" gpurun_out/r04p_bench_8x50M_k21.err | awk 'NR<=12 || NR%10==0' 
grep -c "components: thr" gpurun_out/r04p_bench_8x50M_k21.err
python3 tools/bench_summary.py gpurun_out/r04p_bench_8x50M_k21.json | head -1
