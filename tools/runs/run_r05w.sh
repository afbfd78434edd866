#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_FORCE_DIST=1 timeout -k 5 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --reads 50000000 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05w_bench_50M_rccl_world1.json
python3 -c "
import json; d = json.load(open('gpurun_out/r05w_bench_50M_rccl_world1.json')); print(d['ms_per_step'], d['comm'], d['stage_ms_per_step'])"
