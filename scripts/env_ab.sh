#!/bin/bash
# Same-box A/B of one environment switch on the whole parity-mode network (run on the GPU box through gpurun):
#   scripts/env_ab.sh QMRI_D4 0 1 0 1   -> per-layer kernel times of one 160-slice forward for each value, alternating
# (scripts/c4_ab.sh generalised; output under gpurun_out/ab_<VAR>/)
VAR=$1; shift
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_$VAR
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for v in "$@"; do
  n=$((n+1))
  env $VAR=$v timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$n -o u -- python $GRAFT_REPO_ROOT/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 3 ${UNET_ARGS:-} > $OUT/log$n.txt 2>&1
  f=$(find $OUT/t$n -name "*kernel_trace.csv" | head -1)
  echo "== $VAR=$v (run $n)"; tail -1 $OUT/log$n.txt; python $GRAFT_REPO_ROOT/scripts/unet_trace.py $f 160 | cut -c1-120 | tee $OUT/layers${n}_$v.txt
  rm -rf $OUT/t$n
done
