#!/bin/bash
# A/B of two BUILDS of the library on the parity-mode UNet (run on the GPU box): per-layer times under rocprofv3.
#   scripts/unet_lib_ab.sh prod:dosma_amd/libqmri_hip.so nosat:dosma_amd/libqmri_hip_nosat.so ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lab
i=0
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:}; i=$((i+1))
  DOSMA_AMD_LIB=$lib timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lab/t$i -o u -- python scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 2 > gpurun_out/lab/log$i.txt 2>&1
  f=$(find gpurun_out/lab/t$i -name "*kernel_trace.csv" | head -1)
  echo "== $name"; python scripts/unet_layers.py $f 160 | cut -c1-44
done
