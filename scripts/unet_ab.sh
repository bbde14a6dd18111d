#!/bin/bash
# A/B of an environment switch on the UNet kernels (run on the GPU box through gpurun):
#   scripts/unet_ab.sh QMRI_CONV_W8 0 1           -> per-layer times of scripts/prof_unet.py (bf16) for each value
#   PRECISION=fp16x3 scripts/unet_ab.sh ...       -> the same in the parity mode
VAR=$1; shift
PRECISION=${PRECISION:-bf16}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for v in "$@"; do
  env $VAR=$v timeout -k 5 180 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab/t$v -o u -- python $GRAFT_REPO_ROOT/scripts/prof_unet.py --precision $PRECISION --reps 2 ${UNET_ARGS:-} > $GRAFT_REPO_ROOT/gpurun_out/ab/log$v.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/ab/t$v -name "*kernel_trace.csv" | head -1)
  echo "== $VAR=$v ($PRECISION)"; env $VAR=$v python $GRAFT_REPO_ROOT/scripts/unet_layers.py $f ${UNET_B:-32} | cut -c1-60
done
