#!/bin/bash
# Same-box A/B of library builds (dosma_amd/libqmri_hip<suffix>.so, built with dosma_amd.build.build_variant) on the whole
# parity-mode forward, per layer: scripts/lib_layers_ab.sh "" _nt _sc1 "" _nt _sc1   (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/lib_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for v in "$@"; do
  n=$((n+1))
  env DOSMA_AMD_LIB=$GRAFT_REPO_ROOT/dosma_amd/libqmri_hip$v.so timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$n -o u -- python $GRAFT_REPO_ROOT/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 3 ${UNET_ARGS:-} > $OUT/log$n.txt 2>&1
  f=$(find $OUT/t$n -name "*kernel_trace.csv" | head -1)
  echo "== lib '$v' (run $n)"; tail -1 $OUT/log$n.txt; python $GRAFT_REPO_ROOT/scripts/unet_trace.py $f 160 | cut -c1-120 | tee $OUT/layers${n}$v.txt
  rm -rf $OUT/t$n
done
