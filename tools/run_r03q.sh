cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MF_FUZZ_SCALE=25 timeout -k 5 700 python3 tools/fuzz.py 600 3 > gpurun_out/r03q_fuzz_large.txt 2>&1
timeout -k 5 400 python3 tools/fuzz.py 300 4 > gpurun_out/r03q_fuzz_small.txt 2>&1
timeout -k 5 400 python3 tools/fuzz_cli.py 300 5 > gpurun_out/r03q_fuzz_cli.txt 2>&1
timeout -k 5 300 python3 tools/fuzz_files.py 200 6 > gpurun_out/r03q_fuzz_files.txt 2>&1
tail -2 gpurun_out/r03q_fuzz_large.txt gpurun_out/r03q_fuzz_small.txt gpurun_out/r03q_fuzz_cli.txt gpurun_out/r03q_fuzz_files.txt
