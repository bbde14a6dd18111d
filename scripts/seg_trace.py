import csv, sys, glob
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r["Direction"] ))
ev.sort()
# last H2D of >= 90 MB marks the start of the last call
starts = [i for i, e in enumerate(ev) if e[2].startswith("M") and "HOST_TO_DEVICE" in e[2] and e[1] - e[0] > 500000]
i0 = starts[-1]
t0 = ev[i0][0]
last = None
for s, e, n in ev[i0:]:
    if n.startswith("K qmri::conv") or "c1_split" in n or "maxpool" in n or "conv_" in n:
        if last is None: print(f"{(s - t0)/1e6:8.3f} ms  network starts"); 
        last = e
        continue
    if last is not None: print(f"{(last - t0)/1e6:8.3f} ms  network ends"); last = None
    print(f"{(s - t0)/1e6:8.3f} ms  +{(e - s)/1e6:7.3f}  {n}")
