#!/bin/bash
# same-box A/B of two builds of the library on the whole parity-mode forward: scripts/ab_lib.sh <suffix_a> <suffix_b> [reps] [kernel regex]
# (dosma_amd/libqmri_hip<suffix>.so; "" = the product library): sum of the dispatches matching the regex (default conv_c4_kernel) + total
cd $GRAFT_REPO_ROOT
PAT=${4:-c4_kernel}
for i in $(seq 1 ${3:-2}); do for v in "$1" "$2"; do
  echo -n "lib$v: "; DOSMA_AMD_LIB=$GRAFT_REPO_ROOT/dosma_amd/libqmri_hip$v.so bash scripts/c4_ab.sh 1 2>&1 | grep -E "total|$PAT" | awk '{if ($1!="total") s+=$2; else t=$2} END{print "matching layers", s, "us   forward", t, "us"}'
done; done
