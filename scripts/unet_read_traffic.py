"""Per-layer HBM-side read traffic of one parity-mode forward (160 slices of 384 x 384) against the layer's algorithmic input bytes:
FETCH_SIZE (KB, doubled per the gfx950 correction of MI355X_MICROARCH.md) of the last forward in a rocprofv3 --pmc FETCH_SIZE
collection of scripts/prof_unet.py (scripts/collect_profile.sh writes it to gpurun_out/<tag>/unet_fetch).

    python scripts/unet_read_traffic.py gpurun_out/<tag>/unet_fetch [gpurun_out/<tag>/unet_write] > profiles/<tag>_unet_reads_by_layer.txt   (committed: r04g_…)

With the WRITE_SIZE pass as second argument: also bytes written and the layer's HBM-side bandwidth (kernel time of the FETCH pass).
"""
import collections, csv, glob, re, sys



def load(d, counter):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            disp[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]) * 1024, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    items = list(disp.values())
    return items[[i for i, it in enumerate(items) if "enc0" in it[0]][-1]:]


items = load(sys.argv[1], "FETCH_SIZE")
writes = load(sys.argv[2], "WRITE_SIZE") if len(sys.argv) > 2 else None
start = 0
S = 160
# (layer, level size, input channels incl. the concatenated skip, channel blocks the kernel works in)
layers = [("down0", 384, 1), ("down1.conv1", 192, 32), ("down1.conv2", 192, 64), ("down2.conv1", 96, 64), ("down2.conv2", 96, 128),
          ("down3.conv1", 48, 128), ("down3.conv2", 48, 256), ("down3.pool", 48, 256), ("down4.conv1", 24, 256), ("down4.conv2", 24, 512),
          ("down4.pool", 24, 512), ("down5.conv1", 12, 512), ("down5.conv2", 12, 1024), ("up4.deconv", 12, 1024), ("up4.conv1", 24, 1024),
          ("up4.conv2", 24, 512), ("up3.deconv", 24, 512), ("up3.conv1", 48, 512), ("up3.conv2", 48, 256), ("up2.deconv", 48, 256),
          ("up2.conv1", 96, 256), ("up2.conv2", 96, 128), ("up1.deconv", 96, 128), ("up1.conv1", 192, 128), ("up1.conv2", 192, 64),
          ("up0.deconv", 192, 64), ("up0.conv1", 384, 64), ("up0.conv2", 384, 32)]
print("# reads per layer of one 160-slice forward: FETCH_SIZE x 2 (gfx950 correction) vs the layer's input tensor (4 B per value)")
tf = ta = 0.0
for k, ((name, H, cin), (kern, v, us)) in enumerate(zip(layers, items[start:])):
    alg = S * H * H * cin * 4
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", kern)
    short = (m.group(1) if m else kern)[:34]
    bw = ""
    if writes:
        w = writes[k][1]
        bw = f"   written {w / 1e9:5.2f} GB   {us:6.0f} us -> {(2 * v + w) / us / 1e6:5.2f} TB/s"
    print(f"{name:12s} {short:34s} read {2 * v / 1e9:6.2f} GB   input {alg / 1e9:6.2f} GB   x {2 * v / alg:5.2f}   excess {(2 * v - alg) / 1e9:5.2f} GB{bw}")
    tf += 2 * v
    ta += alg
print(f"total        read {tf / 1e9:.2f} GB   inputs {ta / 1e9:.2f} GB   x {tf / ta:.2f}")
