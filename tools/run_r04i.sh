set -x
mkdir -p gpurun_out
( time ./metafast.sh --version ) 2>&1 | tail -5
( time ./metafast.sh --version ) 2>&1 | tail -5
( time LD_DEBUG=statistics ./metafast_amd/cli/metafast --version ) 2>&1 | grep -i "total startup\|relocation\|real\|load" | head
python3 tools/cli_rate.py 2 20000000 > gpurun_out/r04i_cli_rate.txt 2>&1
tail -45 gpurun_out/r04i_cli_rate.txt
python -m pytest tests/test_files_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r04i_tests.txt 2>&1
tail -5 gpurun_out/r04i_tests.txt
