#!/bin/bash
# PMC of dess_t2_kernel on 8 volumes per launch (run on the GPU box): why 3.3 TB/s and not more -> gpurun_out/pmc_dess/summary.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_dess
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o d -- python $R/scripts/prof_dess.py 10 > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o d -- python $R/scripts/prof_dess.py 3 > $OUT/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o d -- python $R/scripts/prof_dess.py 3 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o d -- python $R/scripts/prof_dess.py 3 > $OUT/write.log 2>&1
python - <<PY > $OUT/summary.txt
import csv, glob
def last(d, name="dess_t2_kernel"):
    rows = [r for r in csv.DictReader(open(glob.glob("$OUT/%s/*counter_collection.csv" % d)[0])) if name in r["Kernel_Name"]]
    i = max(int(r["Dispatch_Id"]) for r in rows)
    return {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == i}
st = [r for r in csv.DictReader(open(glob.glob("$OUT/stats/*kernel_stats.csv")[0])) if "dess_t2_kernel" in r["Name"]][0]
ms = float(st["AverageNs"]) / 1e6
n = 8 * 384 * 384 * 160
sq, fe, wr = last("sq"), last("fetch"), last("write")
cyc = sq["SQ_BUSY_CYCLES"] / 32
print("# scripts/pmc_dess.sh: dess_t2_kernel<float>, 8 volumes of 384x384x160 per launch (f32 echoes -> f64 map, 16 B per voxel)")
print(f"rocprof average {ms:.3f} ms over {st['Calls']} launches -> {16.0 * n / ms / 1e6:.0f} GB/s algorithmic = {16.0 * n / ms / 1e6 / 8000:.3f} of 8 TB/s")
print(f"FETCH_SIZE x 2 (gfx950 correction) {2 * fe['FETCH_SIZE'] * 1024 / 1e9:.3f} GB (algorithmic {8.0 * n / 1e9:.3f}), WRITE_SIZE {wr['WRITE_SIZE'] * 1024 / 1e9:.3f} GB (algorithmic {8.0 * n / 1e9:.3f})")
print(f"VALU busy {sq['SQ_ACTIVE_INST_VALU'] * 4 / (cyc * 1024):.3f} of the SIMD cycles; VALU instructions per voxel {sq['SQ_INSTS_VALU'] * 64 / n:.1f}")
print(f"wave cycles: parked on s_waitcnt {sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']:.3f}, issue-stalled {sq['SQ_WAIT_INST_ANY'] / sq['SQ_WAVE_CYCLES']:.3f}, issuing {sq['SQ_ACTIVE_INST_ANY'] / sq['SQ_WAVE_CYCLES']:.3f}")
print(f"resident waves per SIMD (average) {sq['SQ_WAVE_CYCLES'] * 4 / (cyc * 1024) :.2f}; clock {sq['GRBM_GUI_ACTIVE'] / 8 / (ms * 1e6):.2f} GHz")
PY
cat $OUT/summary.txt
