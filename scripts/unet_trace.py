"""Per-layer table of the LAST U-Net forward in a rocprofv3 --kernel-trace CSV of scripts/prof_unet.py (parity mode).

    python scripts/unet_trace.py <kernel_trace.csv> <batch> [hw]

The op list mirrors forward_batch_parity (unet_engine.hip): c1, conv2 (+ a separate pool where it is not fused), ...,
deconv / conv1 / conv2 per up level.  TF = ALGORITHMIC flops of the layer / kernel time (x3 = MFMA flops issued)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
B = int(sys.argv[2])
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 384
ks = [r for r in rows if "qmri" in r["Kernel_Name"]]
last_whiten = max(i for i, r in enumerate(ks) if "whiten_apply" in r["Kernel_Name"])
ks = ks[last_whiten + 1:]
nf = [32, 64, 128, 256, 512, 1024]


def s3_2d(w):
    return w % 32 == 0


ops = []
enc0 = any("enc0_kernel" in r["Kernel_Name"] for r in ks)  # the first block as one kernel (unet_enc0.hip)
for l in range(6):
    h = HW >> l
    cin = 1 if l == 0 else nf[l - 1]
    if l == 0 and enc0:
        ops.append(("down0 (fused)", h * h * 9 * (nf[0] + nf[0] * nf[0])))
        continue
    ops.append((f"down{l}.conv1", h * h * 9 * cin * nf[l]))
    ops.append((f"down{l}.conv2", h * h * 9 * nf[l] * nf[l]))
    if l < 5:
        fused = s3_2d(h) and nf[l] >= 64 or (not s3_2d(h) and h + 2 > 50)
        if not fused:
            ops.append((f"down{l}.pool", 0))
for l in range(4, -1, -1):
    h = HW >> l
    ops.append((f"up{l}.deconv", (h // 2) ** 2 * 9 * nf[l + 1] * nf[l]))
    ops.append((f"up{l}.conv1", h * h * 9 * 2 * nf[l] * nf[l]))
    ops.append((f"up{l}.conv2", h * h * 9 * nf[l] * nf[l]))
assert len(ks) >= len(ops), (len(ks), len(ops))
ks = ks[: len(ops)]
tot = 0.0
for (name, macs), r in zip(ops, ks):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    kn = r["Kernel_Name"]
    short = kn[kn.find("qmri"):][:58]
    tf = 2 * macs * B / us / 1e6
    print(f"{name:14s} {us:8.0f} us {tf:7.1f} TF  grid {r['Grid_Size_X']:>8} vgpr {r['VGPR_Count']:>3} lds {r['LDS_Block_Size']:>6}  {short}")
print(f"total {tot:.0f} us for {B} slices -> {B / tot * 1e6:.0f} slices/s (kernel time only)")
