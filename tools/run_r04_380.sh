set -x
mkdir -p gpurun_out
python3 bench.py --samples-per-gpu 4 --reads 380000000 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04_bench_4x380M.json 2> gpurun_out/r04_bench_4x380M.err
python3 tools/bench_summary.py gpurun_out/r04_bench_4x380M.json | head -1
tail -3 gpurun_out/r04_bench_4x380M.err | cut -c1-300
MF_VARIANTS="file_mmap=0;file_mmap=1;file_mmap=0;file_mmap=1" python3 tools/cli_rate.py 2 20000000 > gpurun_out/r04_cli_rate.txt 2>&1
grep "total\|write_kmers\|write_comp\|MF_OPTIONS" gpurun_out/r04_cli_rate.txt | tail -30
