"""Summarise a rocprofv3 kernel trace of scripts/prof_unet.py: per-layer time and TFLOP/s of the last forward batch."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
k = [r for r in rows if ('qmri::' in r['Kernel_Name'] or '_ZN4qmri' in r['Kernel_Name']) and 'whiten' not in r['Kernel_Name'] and 'sum' not in r['Kernel_Name'] and 'mean_from' not in r['Kernel_Name']]
# one forward batch = 1 c1 + 26 igemm convs... find last head kernel and go back to the previous head
FUSED_C1 = not any('conv3x3_c1' in r['Kernel_Name'] for r in k)
n_ops = 2 * 6 + 5 * 3 - (1 if FUSED_C1 else 0)
# QMRI_DECONV_SPLIT (bit mask of levels): those transposed convolutions are four launches, merged into one row here
import os
split = int(os.environ.get("QMRI_DECONV_SPLIT", "0"), 0)
n_launch = n_ops + 3 * bin(split & 31).count("1")
raw = k[-n_launch:]
nf = [32, 64, 128, 256, 512, 1024]
# expected op list for flops
ops = []
H = 384
for l in range(6):
    h = H >> l
    cin = 1 if l == 0 else nf[l - 1]
    if l == 0 and FUSED_C1:
        ops.append(("down0.conv1+2", h * h * 9 * (cin * nf[l] + nf[l] * nf[l])))
    else:
        ops.append((f"down{l}.conv1", h * h * 9 * cin * nf[l]))
        ops.append((f"down{l}.conv2", h * h * 9 * nf[l] * nf[l]))
for l in range(4, -1, -1):
    h = H >> l
    hin = h // 2
    ops.append((f"up{l}.deconv", hin * hin * 9 * nf[l + 1] * nf[l]))
    ops.append((f"up{l}.conv1", h * h * 9 * 2 * nf[l] * nf[l]))
    ops.append((f"up{l}.conv2", h * h * 9 * nf[l] * nf[l]))
seg = []
it = iter(raw)
for name, macs in ops:
    n = 4 if (name.endswith(".deconv") and split >> int(name[2]) & 1) else 1
    grp = [next(it) for _ in range(n)]
    r = dict(grp[0])
    r['_us'] = sum(int(g['End_Timestamp']) - int(g['Start_Timestamp']) for g in grp) / 1e3
    seg.append(r)
assert len(ops) == len(seg), (len(ops), len(seg))
tot = 0
for (name, macs), r in zip(ops, seg):
    us = r['_us']
    tot += us
    tf = 2 * macs * B / us / 1e6 if macs else 0
    print(f"{name:18s} {us:8.0f} us  {tf:7.1f} TF  grid {r['Grid_Size_X']:>9}x{r['Grid_Size_Y']} lds {r['LDS_Block_Size']} vgpr {r['VGPR_Count']}+{r['Accum_VGPR_Count']}")
print("total us", tot, "-> slices/s", B / tot * 1e6)
