#!/bin/bash
# PMC of the MFMA kernels of the network (conv_s3 / conv_c4 / deconv_d4 / enc0 / mid0 / out0) in one precision mode (QMRI_C4=0|1|2 in the environment picks the kernel) (run on the GPU box): matrix-pipe busy, waits, LDS activity, clock.
#   scripts/pmc_unet_mode.sh bf16|fp16x3
MODE=${1:-bf16}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$MODE
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT -o u -- python $R/scripts/prof_unet.py --precision $MODE --slices 160 --batch 160 --reps 2 > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
KERNELS = ("conv_s3_kernel", "conv_c4_kernel", "deconv_d4_kernel", "enc0_kernel", "mid0_kernel", "out0_kernel")
rows = [r for r in csv.DictReader(open(glob.glob("$OUT/*counter_collection.csv")[0])) if any(k in r["Kernel_Name"] for k in KERNELS)]
tr = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0]))}
by = collections.defaultdict(collections.Counter)
ids = collections.defaultdict(set)
for r in rows:
    kn = r["Kernel_Name"]
    k = (("c4" + kn.split("conv_c4_kernel")[1][:7]) if "conv_c4_kernel" in kn else ("d4" + kn.split("deconv_d4_kernel")[1][:7]) if "deconv_d4_kernel" in kn
         else kn.split("conv_s3_kernel")[1][:24] if "conv_s3_kernel" in kn else [q for q in KERNELS if q in kn][0])
    by[k][r["Counter_Name"]] += float(r["Counter_Value"]); ids[k].add(r["Dispatch_Id"])
for k, a in by.items():
    cyc = a["SQ_BUSY_CYCLES"] / 32
    ns = sum(tr.get(i, 0) for i in ids[k])
    print(f"{k:26s} n {len(ids[k]):3d}  MfmaUtil {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}  wait_any {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.3f}  wait_inst {a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES']:.3f}  lds_active/cycle/CU {a['SQ_LDS_IDX_ACTIVE'] / (cyc * 256):.3f}  bank_conflict/active {a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1):.3f}  clock GHz {a['GRBM_GUI_ACTIVE'] / 8 / max(ns, 1):.2f}  mfma/us {a['SQ_INSTS_MFMA'] / max(ns, 1) * 1e3:.0f}")
PY
