set -x
mkdir -p gpurun_out
python3 tools/cli_rate.py 2 20000000 > gpurun_out/r04g_cli_rate.txt 2>&1
cat gpurun_out/r04g_cli_rate.txt | tail -80
