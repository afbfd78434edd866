set -x
mkdir -p gpurun_out
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04v_bench_100M.json 2> gpurun_out/r04v_bench_100M.err
python3 tools/bench_summary.py gpurun_out/r04v_bench_100M.json | grep "value\|k_ut_flags\|k_cc_adj" | cut -c1-200
python3 tools/wide_rate.py 200000000 63 > gpurun_out/r04v_wide_200M_k63.json 2> gpurun_out/r04v_wide.err
cat gpurun_out/r04v_wide_200M_k63.json | cut -c1-900; tail -3 gpurun_out/r04v_wide.err
