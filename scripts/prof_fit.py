"""Run only the fit kernel on a cfg2-like slab (for rocprofv3 PMC passes / quick A-B timing)."""
import argparse, ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 23)
ap.add_argument("--recipe", default="A")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--so", default=None)
args = ap.parse_args()
if args.so:
    L._SO = args.so
lib = L.load()
dev = torch.device("cuda", 0)
y = bench.make_volume(torch, dev, 20260928)[:, : args.n].contiguous()
n = y.shape[1]
popt = torch.empty((n, 2), dtype=torch.float32, device=dev)
r2 = torch.empty(n, dtype=torch.float32, device=dev)
a = bench.make_args(L, y, popt, r2, torch.cuda.current_stream().cuda_stream, args.recipe)
ts = []
for i in range(args.reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print(f"recipe {args.recipe} n={n} best {min(ts)*1e3:.3f} ms  {n/min(ts)/1e6:.1f} Mvox/s  all {[round(t*1e3,2) for t in ts]}")
