#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests/test_round5_gpu.py::test_device_parser_steps_back tests/test_round5_gpu.py::test_device_parser_fasta_and_fastq tests/test_distributed_gpu.py -q -x > gpurun_out/r05e_tests.txt 2>&1; tail -15 gpurun_out/r05e_tests.txt
timeout -k 5 300 python3 tools/upload_rate.py 16000000 > gpurun_out/r05e_upload_rate.txt 2>&1; cat gpurun_out/r05e_upload_rate.txt
MF_SIM_SHARDED_ONLY=1 timeout -k 5 900 python3 tools/sim_union.py 8 50000000 > gpurun_out/r05e_sim_union_8x50M.txt 2>&1; tail -14 gpurun_out/r05e_sim_union_8x50M.txt | cut -c1-1500
