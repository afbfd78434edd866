set -x
mkdir -p gpurun_out
MF_OPTIONS=skm_pilot=0 python3 bench.py --samples-per-gpu 4 --reads 380000000 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04_bench_4x380M_nopilot.json 2> /dev/null
python3 tools/bench_summary.py gpurun_out/r04_bench_4x380M_nopilot.json | head -1
MF_OPTIONS=verbose=1 python3 bench.py --samples-per-gpu 4 --reads 380000000 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04_bench_4x380M_v.json 2> gpurun_out/r04_bench_4x380M_v.err
python3 tools/bench_summary.py gpurun_out/r04_bench_4x380M_v.json | head -1
grep -c "new region" gpurun_out/r04_bench_4x380M_v.err; grep "arena\|do not fit\|no room" gpurun_out/r04_bench_4x380M_v.err | tail -30 | cut -c1-200
