cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 3000 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16 > gpurun_out/r03r_tests.txt
cat gpurun_out/r03r_tests.txt
