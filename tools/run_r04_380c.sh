set -x
mkdir -p gpurun_out
python3 bench.py --samples-per-gpu 4 --reads 380000000 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04_bench_4x380M_rec.json 2> /dev/null
python3 tools/bench_summary.py gpurun_out/r04_bench_4x380M_rec.json | grep "value\|k_skm_count " | cut -c1-200
python3 bench.py --samples-per-gpu 8 --reads 200000000 -k 21 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04_bench_8x200M_rec.json 2> /dev/null
python3 tools/bench_summary.py gpurun_out/r04_bench_8x200M_rec.json | grep "value\|k_skm_count \|k_skm_split " | cut -c1-200
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04_bench_100M_rec.json 2> /dev/null
python3 tools/bench_summary.py gpurun_out/r04_bench_100M_rec.json | grep "value\|k_skm_count " | cut -c1-200
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --genome-scale 16000000 > gpurun_out/r04_bench_5x_rec.json 2> /dev/null
python3 tools/bench_summary.py gpurun_out/r04_bench_5x_rec.json | grep "value\|k_skm_count " | cut -c1-200
