#!/bin/bash
# the whole GPU suite at HEAD, then the other shapes of the hash-count kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 3300 python3 -m pytest tests -m gpu -q -x > gpurun_out/r05f_gpu_tests.txt 2>&1; tail -8 gpurun_out/r05f_gpu_tests.txt
bash tools/other_shapes.sh r05f
