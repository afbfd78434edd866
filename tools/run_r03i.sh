cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1200 python3 bench.py --no-cpu-baseline --samples-per-gpu 4 --reads 380000000 --steps 1 --warmup 0 > gpurun_out/r03i_bench_4x380M.json 2> gpurun_out/r03i_bench_4x380M.err
tail -c 1500 gpurun_out/r03i_bench_4x380M.err; python3 -c "
import json
d=json.load(open('gpurun_out/r03i_bench_4x380M.json')); print(d['ms_per_step'], d['value'], d['stage_ms_per_step'], d['stats'])"
