cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r03h_tests.txt


timeout -k 5 900 python3 tools/sim_union.py 8 50000000 > gpurun_out/r03h_sim_union.txt 2>&1
cat gpurun_out/r03h_tests.txt; tail -5 gpurun_out/r03h_sim_union.txt; python3 -c "
import json
for f in ():
    d=json.load(open(f)); print(d['ms_per_step'], d['stage_ms_per_step'], d['comm']); print({k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
