set -x
mkdir -p gpurun_out
python -m pytest tests/test_files_gpu.py tests/test_count_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r04y_tests.txt 2>&1
tail -3 gpurun_out/r04y_tests.txt
timeout 200 python3 tools/fuzz_files.py 60 5 > gpurun_out/r04y_fuzz_files.txt 2>&1; tail -1 gpurun_out/r04y_fuzz_files.txt
python3 tools/file_path_rate.py 8000000 2>&1 | grep "file path"
MF_IO_TIMING=1 python3 bench.py --steps 5 --warmup 2 > gpurun_out/r04y_bench_100M.json 2> gpurun_out/r04y_bench_100M.err
python3 tools/bench_summary.py gpurun_out/r04y_bench_100M.json | grep "value\|k_skm_count \|roofline" | cut -c1-260
python3 -c "
import json; d=json.load(open('gpurun_out/r04y_bench_100M.json')); print(d['end_to_end']); print(d['cli']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
grep "count_reads\|driver" gpurun_out/r04y_bench_100M.err | tail -8
