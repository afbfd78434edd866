cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_OPTIONS=nbr_batch=2 timeout -k 5 600 python3 bench.py --no-cpu-baseline > gpurun_out/r03j_bench_batch2.json 2> gpurun_out/r03j_bench_batch2.err
timeout -k 5 1200 python3 bench.py --no-cpu-baseline --samples-per-gpu 4 --reads 380000000 --steps 1 --warmup 1 > gpurun_out/r03j_bench_4x380M.json 2> gpurun_out/r03j_bench_4x380M.err
python3 -c "
import json
d=json.load(open('gpurun_out/r03j_bench_batch2.json')); print(d['ms_per_step'], d['stage_ms_per_step']); print({k:round(v['ms_per_step'],2) for k,v in d['kernels'].items() if 'flags' in k or 'adjac' in k})
d=json.load(open('gpurun_out/r03j_bench_4x380M.json')); print(d['ms_per_step'], d['value'], d['stage_ms_per_step']); print({k:round(v['ms_per_step'],1) for k,v in d['kernels'].items()})"
