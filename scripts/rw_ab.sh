cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rw
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q 2>&1 | tail -8
for rw in 1; do
  QMRI_DECONV_BN=32 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rw/t$rw -o u -- python $GRAFT_REPO_ROOT/scripts/prof_unet.py --precision bf16 --reps 2 > $GRAFT_REPO_ROOT/gpurun_out/rw/log$rw.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/rw/t$rw -name "*kernel_trace.csv" | head -1)
  echo "== RW=$rw"; python $GRAFT_REPO_ROOT/scripts/unet_layers.py $f 32 | grep -E "down0|down1|up1|up0|total"
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/rw/log$rw.txt
done
