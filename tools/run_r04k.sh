set -x
mkdir -p gpurun_out
MF_VARIANTS="verbose=0;stream_piece_bytes=4194304;stream_piece_bytes=2097152;verbose=0" python3 tools/cli_rate.py 2 20000000 > gpurun_out/r04k_cli_rate.txt 2>&1
grep "total\|\[mf\]\|MF_OPTIONS" gpurun_out/r04k_cli_rate.txt | tail -60
