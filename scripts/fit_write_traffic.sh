#!/bin/bash
# Run on the GPU box: WRITE_SIZE / FETCH_SIZE of the headline fit kernel (one launch each) + its time.
# usage: scripts/fit_write_traffic.sh <outdir>
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-fitw}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cfg5 --no-parity --no-unet --recipes A"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o w -- $B > $OUT/w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o f -- $B > $OUT/f.log 2>&1
python - <<PY
import csv, glob
for d, c in (("w", "WRITE_SIZE"), ("f", "FETCH_SIZE")):
    rows = [r for r in csv.DictReader(open(glob.glob("$OUT/%s/*counter_collection.csv" % d)[0])) if "monoexp_lm_kernel" in r["Kernel_Name"] and ("Li8ELb1EfLb0" in r["Kernel_Name"] or "<8, true, float, false>" in r["Kernel_Name"] or "<8, true, float>" in r["Kernel_Name"])]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    v = sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last and r["Counter_Name"] == c)
    print(c, "MB per launch:", v * 1024 / 1e6)
PY
python $R/bench.py --no-unet --no-cfg5 --no-cpu-baseline --no-parity | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'])"
