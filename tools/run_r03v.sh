cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_distributed_gpu.py tests/test_shapes_gpu.py::test_eight_samples_one_gpu_against_the_oracle -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r03v_tests.txt
timeout -k 5 900 python3 tools/sim_union.py 8 50000000 > gpurun_out/r03v_sim_union.txt 2>&1
cat gpurun_out/r03v_tests.txt; tail -3 gpurun_out/r03v_sim_union.txt | cut -c1-600
