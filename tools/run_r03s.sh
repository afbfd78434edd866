cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_count_gpu.py -x -q -m gpu -k "identical" 2>&1 | grep -v "^$" | tail -40
