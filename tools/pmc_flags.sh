#!/bin/bash
# where the waves of k_ut_flags_part spend their cycles: three rocprofv3 --pmc passes over a 100 M-read step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d gpurun_out/pmc_flags_$i -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > gpurun_out/pmc_flags_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
for f in glob.glob('gpurun_out/pmc_flags_$i/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_ut_flags_part' in r['Kernel_Name'] and 'Li1E' in r['Kernel_Name'] or ('k_ut_flags_part<1' in r['Kernel_Name']):
            acc[r['Counter_Name']]+=float(r['Counter_Value'])
    print($i, dict(acc))
PY
done
