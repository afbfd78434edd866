#!/bin/bash
# round 5, second GPU call: the new tests (device parser, --devices, --selected, CAMI parameters), the file fuzzer, the file path's rate
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 2400 python3 -m pytest tests/test_round5_gpu.py -q -x --deselect tests/test_round5_gpu.py::test_pipeline_against_the_oracle_other_k > gpurun_out/r05b_tests.txt 2>&1; tail -25 gpurun_out/r05b_tests.txt
timeout -k 5 200 python3 tools/fuzz_files.py 90 5 > gpurun_out/r05b_fuzz_files.txt 2>&1; tail -5 gpurun_out/r05b_fuzz_files.txt
MF_IO_TIMING=1 timeout -k 5 300 python3 tools/file_path_rate.py 16000000 > gpurun_out/r05b_file_path_rate.txt 2>&1; grep -v "^\[mf\] arena" gpurun_out/r05b_file_path_rate.txt | tail -20
