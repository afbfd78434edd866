cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_count_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r03a_tests.txt
timeout -k 5 600 python3 tools/count_ab.py 100000000 31 6144:count_group=1,6144,12288 > gpurun_out/r03a_count_ab.txt 2>&1
tail -20 gpurun_out/r03a_tests.txt gpurun_out/r03a_count_ab.txt
