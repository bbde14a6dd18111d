# Do the top-level kernels get faster when their inputs are infinity-cache resident?  The whole network in sub-batches of 8 / 4 slices
# against one batch of 160 (run on the GPU box): per-kernel time per 160-slice forward.  Answer (round 4): no.
cd /tmp && export TMPDIR=/tmp
for b in 160 8 4; do
  rm -rf /tmp/mp$b; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp$b -o u -- python $GRAFT_REPO_ROOT/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch $b --reps 2 > /tmp/mp$b.log 2>&1
  f=$(find /tmp/mp$b -name "*kernel_stats.csv" | head -1)
  echo "== batch $b"; tail -1 /tmp/mp$b.log
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('mid0','out0','enc0','conv_s3_kernel<32, false, true')):
        print(f"  {n[:70]:70s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6/2:8.3f} ms per forward")
PY
done
