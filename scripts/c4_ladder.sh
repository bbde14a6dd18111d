#!/bin/bash
# conv_c4_kernel with one ingredient removed at a time (experiment build: python -m dosma_amd.build --variant c4x -DQMRI_C4_EXPERIMENTS;
# results wrong, timing informative): per-layer times of one 160-slice forward with QMRI_C4=2.
#   QMRI_C4_DBG = 1 no epilogue | 2 halo sources computed once | 4 no MFMAs | 8 no LDS reads | 16 no DMA requests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4lad
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  env DOSMA_AMD_LIB=$GRAFT_REPO_ROOT/dosma_amd/libqmri_hip_c4x.so QMRI_C4=2 QMRI_C4_DBG=$v timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c4lad/t$v -o u -- python $GRAFT_REPO_ROOT/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 2 > $GRAFT_REPO_ROOT/gpurun_out/c4lad/log$v.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/c4lad/t$v -name "*kernel_trace.csv" | head -1)
  python $GRAFT_REPO_ROOT/scripts/unet_trace.py $f 160 > $GRAFT_REPO_ROOT/gpurun_out/c4lad/layers$v.txt
  echo "dbg $v: $(grep 'conv_c4_kernel<.*, 4>' $GRAFT_REPO_ROOT/gpurun_out/c4lad/layers$v.txt | awk '{s+=$2} END{printf "c4x128 layers %d us", s}')  $(grep -E 'down2.conv1|down3.conv1|up3.conv1|down2.conv2' $GRAFT_REPO_ROOT/gpurun_out/c4lad/layers$v.txt | awk '{printf "%s %d  ", $1, $2}')"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/c4lad/t$v
done
