#!/bin/bash
# Run on the GPU box (through gpurun): bench + rocprofv3 kernel stats + HBM PMC passes -> gpurun_out/<tag>/
# usage: scripts/collect_profile.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cfg5 --no-parity --no-unet --recipes A > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cfg5 --no-parity --no-unet --recipes A > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cfg5 --no-parity --no-unet --recipes A > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS_F64 SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cfg5 --no-parity --no-unet --recipes A > $OUT/pmc_sq.log 2>&1
tail -c 600 $OUT/bench.json
# ---- UNet2D leg: per-layer kernel trace + MFMA counters of scripts/prof_unet.py in the parity mode (fp16x3, one 160-slice volume) ----
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/unet_stats -o u -- python $R/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 2 > $OUT/unet_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/unet_pmc -o u -- python $R/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 2 > $OUT/unet_pmc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/unet_fetch -o u -- python $R/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 2 > $OUT/unet_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/unet_write -o u -- python $R/scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160 --reps 2 > $OUT/unet_write.log 2>&1
python $R/bench.py --print-kernel-hash > $OUT/kernel_hash.txt
python $R/bench.py --print-unet-hash > $OUT/unet_hash.txt
# ---- per-kernel-family PMC of the parity mode (MfmaUtil, waits, LDS, clock, MFMAs per us) and the deconv_d4 / conv_s3 A/B of the transposed convolutions on this box ----
bash $R/scripts/pmc_unet_mode.sh fp16x3 > $OUT/unet_pmc_by_kernel.txt 2>&1
bash $R/scripts/env_ab.sh QMRI_D4 0 1 0 1 > $OUT/d4_ab.txt 2>&1
(cd $R && python -m dosma_amd.build --clean > /dev/null)  # experiment variants / probe binaries do not travel with the next push
