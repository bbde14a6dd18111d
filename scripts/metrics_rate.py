"""QuantitativeValue.to_metrics on a 512 x 512 x 160 float64 map with a 4-label mask: GPU route vs the host (numpy) route."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dosma_amd as dm
from dosma_amd.quant_vals import T2
rng = np.random.default_rng(0)
shape = (512, 512, 160)
v = np.around(rng.uniform(0, 100, shape), 1)
lab = (rng.integers(0, 50, shape) // 10 % 5).astype(np.uint8)
qv = T2(dm.MedicalVolume(v, np.eye(4)))
mask = dm.MedicalVolume(lab, np.eye(4))
labels = {1: "fc", 2: "tc", 3: "pc", 4: "men"}
for name, kw in (("gpu ", {}), ("host", {"fns": {"n": lambda a: a.size}})):
    ts = []
    for _ in range(3):
        t = time.perf_counter(); df = qv.to_metrics(mask, labels, bounds=(0, 100), **kw); ts.append(time.perf_counter() - t)
    print(f"{name}: {min(ts)*1e3:8.1f} ms   (all {[round(x*1e3) for x in ts]})")
print(df)
