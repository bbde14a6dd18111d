"""Turn a gpurun_out/<tag>/ directory (scripts/collect_profile.sh) into the committed summaries under
profiles/: <tag>_kernel_stats.csv (rocprofv3 --stats), <tag>_bench.json, <tag>_counters.json
(HBM traffic with the gfx950 FETCH_SIZE correction + SQ counters)."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "stats", "k_kernel_stats.csv"), f"profiles/{tag}_kernel_stats.csv")
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(f"profiles/{tag}_bench.json", "w"), indent=1)


HEADLINE = ("monoexp_lm_kernel<8, true, float, false>", "monoexp_lm_kernel<8, true, float>", "monoexp_lm_kernelILi8ELb1EfLb0E", "monoexp_lm_kernelILi8ELb1EfEE")  # the bench's headline variant (dense, not LISTED)


def last_dispatch(path, kernel=HEADLINE):
    rows = [r for r in csv.DictReader(open(path)) if any(k in r["Kernel_Name"] for k in kernel)]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    return {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last}, rows[0]


fetch, _ = last_dispatch(glob.glob(os.path.join(src, "pmc_fetch", "*counter_collection.csv"))[0])
write, _ = last_dispatch(glob.glob(os.path.join(src, "pmc_write", "*counter_collection.csv"))[0])
sq, row = last_dispatch(glob.glob(os.path.join(src, "pmc_sq", "*counter_collection.csv"))[0])
n = bench["config"]["voxels_per_gpu_per_step"]
alg = bench["roofline"]["algorithmic_bytes_per_voxel"] * n
rd = 2 * fetch["FETCH_SIZE"] * 1024   # MI355X_MICROARCH.md: FETCH_SIZE counts 1/2 of a wide coalesced read
wr = write["WRITE_SIZE"] * 1024
stats = [r for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")) if any(k in r["Name"] for k in HEADLINE)][0]
out = {
    "tag": tag,
    "kernel": stats["Name"],
    "rocprof_avg_kernel_ms": float(stats["AverageNs"]) / 1e6,
    "bench_hip_event_kernel_ms": bench["roofline"]["kernel_ms"],
    "vgpr": row["VGPR_Count"], "accum_vgpr": row["Accum_VGPR_Count"], "lds_bytes_per_block": row["LDS_Block_Size"],
    "FETCH_SIZE_raw_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_raw_KB": write["WRITE_SIZE"],
    "correction": "FETCH_SIZE doubled (gfx950: reports 1/2 of the bytes of a wide coalesced streaming read, "
                  "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (uncalibrated per the guide)",
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
    "hbm_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": (rd + wr) / alg,
    "sq": sq,
    "lanes_active_per_valu_instr": sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"],
    # SQ_ACTIVE_INST_VALU counts quad-cycles of VALU execution summed over SIMDs; SQ_BUSY_CYCLES is summed over the
    # 32 shader engines (8 XCD x 4) -> kernel duration in clocks = SQ_BUSY_CYCLES / 32; 1024 SIMDs
    "valu_busy_frac": sq["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * sq["SQ_BUSY_CYCLES"] / 32.0),
}
json.dump(out, open(f"profiles/{tag}_counters.json", "w"), indent=1)
out["valu_lane_instr_per_launch"] = sq["SQ_INSTS_VALU"] * out["lanes_active_per_valu_instr"]
json.dump(out, open(f"profiles/{tag}_counters.json", "w"), indent=1)
rnd = tag[:3]  # "r02"
sha = open(os.path.join(src, "kernel_hash.txt")).read().strip() if os.path.exists(os.path.join(src, "kernel_hash.txt")) else None
json.dump({"hbm_bytes_per_launch": rd + wr, "valu_lane_instr_per_launch": out["valu_lane_instr_per_launch"],
           "source": f"profiles/{tag}_counters.json", "kernel_source_sha1": sha,
           "pmc": {"valu_busy_frac": out["valu_busy_frac"], "lanes_active_per_valu_instr": out["lanes_active_per_valu_instr"],
                   "traffic_over_algorithmic": out["traffic_over_algorithmic"]}},
          open(f"profiles/{rnd}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))

# ---- UNet2D leg (parity mode fp16x3, one 160-slice volume per forward) ----
import collections, subprocess
ut = os.path.join(src, "unet_stats", "u_kernel_trace.csv")
if os.path.exists(ut):
    shutil.copy(os.path.join(src, "unet_stats", "u_kernel_stats.csv"), f"profiles/{tag}_unet_kernel_stats.csv")
    layers = subprocess.check_output([sys.executable, "scripts/unet_trace.py", ut, "160"]).decode()
    open(f"profiles/{tag}_unet_layers.txt", "w").write(
        "# rocprofv3 --kernel-trace of scripts/prof_unet.py --precision fp16x3 --slices 160 --batch 160: last forward\n"
        "# TF = ALGORITHMIC flops of the layer / kernel time (the parity mode issues 3 MFMAs per product)\n" + layers)
    rows = [r for r in csv.DictReader(open(glob.glob(os.path.join(src, "unet_pmc", "*counter_collection.csv"))[0]))
            if any(k in r["Kernel_Name"] for k in ("conv_s3_kernel", "conv_c4_kernel", "enc0_kernel", "mid0_kernel", "out0_kernel"))]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-25:]  # per forward: enc0 + 17 convolutions + 5 transposed + mid0 + out0
    agg = collections.Counter()
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    cycles = agg["SQ_BUSY_CYCLES"] / 32  # summed over the 32 shader engines
    u = {"tag": tag, "workload": "UNet2D forward, 160 slices of 384x384, parity mode fp16x3: the 25 MFMA-kernel dispatches of one forward "
                                 "(enc0_kernel, 22 x conv_s3_kernel / conv_c4_kernel, mid0_kernel, out0_kernel)",
         "counters": dict(agg),
         "MfmaUtil": agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024),
         "mfma_instructions": agg["SQ_INSTS_MFMA"],
         "mfma_gflop_issued": agg["SQ_INSTS_MFMA"] * 32768 / 1e9,
         "mfma_gflop_algorithmic_x3": 3 * 70.79 * 160,
         "lds_bank_conflict_over_active": agg["SQ_LDS_BANK_CONFLICT"] / max(agg["SQ_LDS_IDX_ACTIVE"], 1.0),
         "wave_wait_frac": agg["SQ_WAIT_ANY"] / max(agg["SQ_WAVE_CYCLES"], 1.0),
         "unet2d_bench": bench.get("unet2d")}
    # HBM traffic of one forward: FETCH_SIZE / WRITE_SIZE (KB) summed over the qmri kernels of the LAST forward
    def last_forward_sum(d, counter):
        f = glob.glob(os.path.join(src, d, "*counter_collection.csv"))
        if not f:
            return None
        rr = [r for r in csv.DictReader(open(f[0])) if "qmri" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        starts = [int(r["Dispatch_Id"]) for r in rr if "whiten_apply" in r["Kernel_Name"]]
        if not starts:
            return None
        return sum(float(r["Counter_Value"]) for r in rr if int(r["Dispatch_Id"]) > max(starts)) * 1024.0
    fe, wr_ = last_forward_sum("unet_fetch", "FETCH_SIZE"), last_forward_sum("unet_write", "WRITE_SIZE")
    if fe is not None and wr_ is not None:
        u["hbm"] = {"FETCH_SIZE_bytes_raw": fe, "WRITE_SIZE_bytes": wr_,
                    "read_bytes_corrected": 2 * fe, "bytes_per_forward": 2 * fe + wr_,
                    "correction": "FETCH_SIZE doubled (gfx950 counts 1/2 of wide coalesced / LDS-DMA reads, MI355X_MICROARCH.md); WRITE_SIZE as reported",
                    "per_slice_MB": (2 * fe + wr_) / 160 / 1e6}
    uh = os.path.join(src, "unet_hash.txt")
    u["kernel_source_sha1"] = open(uh).read().strip() if os.path.exists(uh) else None   # bench.py: traffic_source.stale
    json.dump(u, open(f"profiles/{tag}_unet_counters.json", "w"), indent=1)
    print(json.dumps({k: u[k] for k in ("MfmaUtil", "mfma_gflop_issued", "mfma_gflop_algorithmic_x3", "wave_wait_frac")}))

# ---- round 4: per-kernel-family PMC table and the same-box conv_c4 / conv_s3 A/B ----
for name in ("unet_pmc_by_kernel.txt", "c4_ab.txt", "d4_ab.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        # (the A/B is referenced from the kernel sources as profiles/r04_c4_ab.txt: one per round, the newest collection)
        shutil.copy(f, f"profiles/{rnd}_{name}" if name in ("c4_ab.txt", "d4_ab.txt") else f"profiles/{tag}_{name}")
