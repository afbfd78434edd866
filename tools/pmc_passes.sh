#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <reads>   -> gpurun_out/pmc_<tag>_{a,b,c,d}/ (one rocprofv3 --pmc pass each)
TAG=$1; READS=${2:-20000000}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc_${TAG}_$1 -o p -- python3 tools/prof_count.py $READS 1 > gpurun_out/pmc_${TAG}_$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
if [ "$3" != "sq_only" ]; then run c "FETCH_SIZE"; fi
if [ "$3" != "sq_only" ]; then run d "WRITE_SIZE"; fi
ls gpurun_out/pmc_${TAG}_*/ | head -20
