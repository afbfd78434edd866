set -x
mkdir -p gpurun_out
MF_VARIANTS="host_pinned=1;host_pinned=0;host_pinned=1;host_pinned=0" python3 tools/cli_rate.py 2 20000000 > gpurun_out/r04l_cli_rate.txt 2>&1
grep "total\|\[mf\]\|MF_OPTIONS" gpurun_out/r04l_cli_rate.txt | tail -60
python3 tools/file_path_rate.py 8000000 2>&1 | grep "file path\|count_reads"
