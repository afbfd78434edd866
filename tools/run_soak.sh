cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_FUZZ_SCALE=25 timeout -k 5 420 python3 tools/fuzz.py 400 61 2>&1 | tail -n 1
timeout -k 5 300 python3 tools/fuzz.py 280 62 2>&1 | tail -n 1
timeout -k 5 200 python3 tools/fuzz_cli.py 180 63 2>&1 | tail -n 1
timeout -k 5 600 python -m pytest tests/test_shapes_gpu.py tests/test_config5_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -3
