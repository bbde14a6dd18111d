#!/bin/bash
# deconv_d4_kernel: variant builds and the experiment build with one ingredient removed at a time, same box.
#   python -m dosma_amd.build --variant d4x -DQMRI_D4_EXPERIMENTS   (+ any --variant <name> -D... to compare: pass the names as arguments)
# usage: scripts/d4_ladder.sh [variant ...]            (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
run() {  # run <label> <lib suffix> [env...]
  local label=$1 lib=$2; shift 2
  env DOSMA_AMD_LIB=$R/dosma_amd/libqmri_hip$lib.so "$@" bash scripts/c4_ab.sh 1 2>&1 | grep -E "deconv|total" | awk -v L="$label" '{if ($1=="total") t=$2; else printf "%s %d  ", $1, $2} END{print " | forward", t, "us   <-", L}'
}
for rep in 1 2; do
  run "product" ""
  for v in "$@"; do run "variant $v" _$v; done
  run "conv_s3_kernel (QMRI_D4=0)" "" QMRI_D4=0
done
if [ -f $R/dosma_amd/libqmri_hip_d4x.so ]; then
  for d in ${D4_DBGS:-0 1 16 8 24 4 32}; do run "experiment build, QMRI_D4_DBG=$d" _d4x QMRI_D4_DBG=$d; done
fi
