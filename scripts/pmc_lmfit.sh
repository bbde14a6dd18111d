#!/bin/bash
# Run on the GPU box (through gpurun): SQ counters of the general lmdif kernels (scripts/prof_lmfit.py) -> gpurun_out/<tag>/
TAG=${1:-lmfit}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -o a -- python $R/scripts/prof_lmfit.py --reps 2 > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH --kernel-trace --output-format csv -d $OUT/b -o b -- python $R/scripts/prof_lmfit.py --reps 2 > $OUT/b.log 2>&1
cd $R
for k in lm_pull_kernel; do for f in $(find $OUT -name "*counter_collection.csv"); do python scripts/pmc_table.py $f $k; done; done
tail -3 $OUT/a.log
