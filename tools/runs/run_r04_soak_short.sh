mkdir -p gpurun_out
MF_FUZZ_SCALE=25 timeout 250 python3 tools/fuzz.py 180 601 > gpurun_out/r04c_fuzz_large.txt 2>&1; tail -1 gpurun_out/r04c_fuzz_large.txt
timeout 250 python3 tools/fuzz.py 180 602 > gpurun_out/r04c_fuzz_small.txt 2>&1; tail -1 gpurun_out/r04c_fuzz_small.txt
timeout 250 python3 tools/fuzz_cli.py 180 603 > gpurun_out/r04c_fuzz_cli.txt 2>&1; tail -1 gpurun_out/r04c_fuzz_cli.txt
