#!/bin/bash
# PMC of the MFMA kernels' global stores in the parity-mode forward (run on the GPU box): cycles waves spend issuing VMEM writes, TA FIFO
# back-pressure, per kernel family.   scripts/pmc_unet_stores.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_stores
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT -o u -- python $R/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 2 > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
KERNELS = ("conv_s3_kernel", "conv_c4_kernel", "deconv_d4_kernel", "enc0_kernel", "mid0_kernel", "out0_kernel")
rows = [r for r in csv.DictReader(open(glob.glob("$OUT/*counter_collection.csv")[0])) if any(k in r["Kernel_Name"] for k in KERNELS)]
by = collections.defaultdict(collections.Counter); ids = collections.defaultdict(set)
for r in rows:
    kn = r["Kernel_Name"]
    k = (("c4" + kn.split("conv_c4_kernel")[1][:9]) if "conv_c4_kernel" in kn else ("d4" + kn.split("deconv_d4_kernel")[1][:7]) if "deconv_d4_kernel" in kn else [q for q in KERNELS if q in kn][0])
    by[k][r["Counter_Name"]] += float(r["Counter_Value"]); ids[k].add(r["Dispatch_Id"])
print("# per kernel family, one forward x 2: VMEM-write instructions per wave-cycle budget; SQ_INST_CYCLES_VMEM_WR / SQ_WAVE_CYCLES = share of wave time in store issue")
for k, a in sorted(by.items()):
    wc = a["SQ_WAVE_CYCLES"]
    print(f"{k:18s} n {len(ids[k]):3d}  stores {a['SQ_INSTS_VMEM_WR']:.3e}  cycles_in_store_issue/wave_cycles {a['SQ_INST_CYCLES_VMEM_WR'] / wc:.3f}  (per store {a['SQ_INST_CYCLES_VMEM_WR'] / max(a['SQ_INSTS_VMEM_WR'], 1):.0f} quad-cycles)  "
          f"TA data fifo full/wave_cycles {a['SQ_VMEM_WR_TA_DATA_FIFO_FULL'] / wc:.3f}  addr fifo full {a['SQ_VMEM_TA_ADDR_FIFO_FULL'] / wc:.3f}  cmd fifo full {a['SQ_VMEM_TA_CMD_FIFO_FULL'] / wc:.3f}  wait_lds {a['SQ_WAIT_INST_LDS'] / wc:.3f}")
PY
