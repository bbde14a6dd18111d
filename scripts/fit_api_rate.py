"""End-to-end time of MonoExponentialFit.fit on eight 512 x 512 x 160 float32 MedicalVolumes (host in, host out)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dosma_amd as dm

shape = (512, 512, 160)
rng = np.random.default_rng(0)
x = np.arange(1, 9) * 10.0
t2 = rng.uniform(15, 80, shape).astype(np.float32)
s0 = rng.uniform(300, 1500, shape).astype(np.float32)
vols = []
for t in x:
    v = (s0 * np.exp(-np.float32(t) / t2)).astype(np.float32)
    v += rng.standard_normal(shape, dtype=np.float32) * 18
    v[:150] = 0
    vols.append(dm.MedicalVolume(v, np.eye(4)))
n = np.prod(shape)
# the same volumes as int16 (what DICOM pixel data is; the reference keeps the volumes' dtype, fitting.py:194-196)
vols16 = [dm.MedicalVolume(np.rint(v.volume).astype(np.int16), np.eye(4)) for v in vols]
for name, vv in (("float32", vols), ("int16", vols16)):
    for tc0 in (30.0, "polyfit"):
        f = dm.MonoExponentialFit(tc0=tc0, decimal_precision=3)
        for rep in range(3):
            t0 = time.perf_counter()
            tc, r2 = f.fit(x, vv)
            dt = time.perf_counter() - t0
        print(f"{name}: MonoExponentialFit(tc0={tc0!r}).fit: {dt*1e3:.1f} ms -> {n/dt:.3e} voxel-fits/s end to end")
pr = cProfile.Profile(); pr.enable(); f.fit(x, vols); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
