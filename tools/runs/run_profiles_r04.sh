set -x
mkdir -p gpurun_out
bash tools/refresh_profiles.sh r04ap > gpurun_out/r04ap_refresh.log 2>&1
tail -5 gpurun_out/r04ap_refresh.log
MF_OPTIONS=skm_dedupe=0 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r04ap_bench_100M_dedupe0.json 2> /dev/null
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --genome-scale 16000000 > gpurun_out/r04ap_bench_100M_5fold_depth.json 2> /dev/null
python3 tools/wide_rate.py 200000000 63 > gpurun_out/r04ap_wide_200M_k63.json 2> /dev/null
for f in r04ap_bench_100M r04ap_bench_100M_dedupe0 r04ap_bench_100M_5fold_depth; do python3 tools/bench_summary.py gpurun_out/$f.json | grep "value\|k_skm_count " | cut -c1-220; done
cut -c1-600 gpurun_out/r04ap_wide_200M_k63.json
