#!/bin/bash
# A/B of conv_c4_kernel against conv_s3_kernel<128> on the whole network (run on the GPU box through gpurun):
#   scripts/c4_ab.sh 0 2 1   -> per-layer times of one 160-slice forward (parity mode) for QMRI_C4 = 0 (never) / 2 (wherever it
#   is supported) / 1 (the launcher's cost model), alternating on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4ab
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  env QMRI_C4=$v timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c4ab/t$v -o u -- python $GRAFT_REPO_ROOT/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 3 ${UNET_ARGS:-} > $GRAFT_REPO_ROOT/gpurun_out/c4ab/log$v.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/c4ab/t$v -name "*kernel_trace.csv" | head -1)
  echo "== QMRI_C4=$v"; tail -1 $GRAFT_REPO_ROOT/gpurun_out/c4ab/log$v.txt; python $GRAFT_REPO_ROOT/scripts/unet_trace.py $f 160 | cut -c1-120 | tee $GRAFT_REPO_ROOT/gpurun_out/c4ab/layers$v.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/c4ab/t$v
done
