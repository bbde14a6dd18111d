#!/bin/bash
# Run on the GPU box: per-layer times of the parity-mode forward with one ingredient of conv_s3_kernel removed at a time
# (experiment build: python -m dosma_amd.build --experiments; results are WRONG by construction, timing only)
R=$GRAFT_REPO_ROOT
export DOSMA_AMD_LIB=$R/dosma_amd/libqmri_hip_exp.so
for v in "$@"; do
  OUT=$R/gpurun_out/ladder/d$v; mkdir -p $OUT
  ( cd /tmp; export TMPDIR=/tmp; QMRI_S3_DBG=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o u -- python $R/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 2 > $OUT/log.txt 2>&1 )
  echo "== QMRI_S3_DBG=$v"; python $R/scripts/unet_trace.py $OUT/u_kernel_trace.csv 160 | grep -E "down0.conv2|down1.conv2|down3.conv2|up1.conv1|up0.deconv|up0.conv1|up0.conv2|total" | cut -c1-48
done
