#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 600 python3 -m pytest tests/test_round5_gpu.py -q -x -k "radix" > gpurun_out/r05aa_sort_tests.txt 2>&1; tail -5 gpurun_out/r05aa_sort_tests.txt
timeout -k 5 2400 python3 -m pytest tests/test_files_gpu.py tests/test_pipeline_gpu.py tests/test_wide_gpu.py tests/test_golden_fixture.py tests/test_distributed_gpu.py tests/test_round4_gpu.py -q -x > gpurun_out/r05aa_tests.txt 2>&1; tail -4 gpurun_out/r05aa_tests.txt
MF_IO_TIMING=1 timeout -k 5 300 python3 tools/cli_rate.py 2 20000000 2>&1 | grep "write_kmers\|write_components\|write_fasta\|exit\|total" | tail -12
timeout -k 5 600 python3 tools/wide_rate.py 200000000 63 2>/dev/null | cut -c1-400
