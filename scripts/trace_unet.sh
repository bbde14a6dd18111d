#!/bin/bash
# Run on the GPU box (through gpurun): per-layer kernel trace of one UNet2D forward in a precision mode.
# usage: scripts/trace_unet.sh <tag> [precision] [reps]
TAG=${1:-trace}; PREC=${2:-fp16x3}; REPS=${3:-3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o u -- python $R/scripts/prof_unet.py --precision $PREC --slices 160 --batch 160 --reps $REPS > $OUT/trace.log 2>&1
cd $R
python scripts/unet_trace.py $OUT/u_kernel_trace.csv 160 > $OUT/layers.txt
tail -50 $OUT/layers.txt
