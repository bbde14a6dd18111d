"""The other kernels, repeated: are the fit (device entry, recipes A and B, masked), the general lmdif kernel (bi-exponential), the DESS map and
the region statistics bit-identical from launch to launch?  (Round 6: after conv_c4_kernel's race -- found only because a bench leg compared two
runs -- every kernel family gets a many-launch repeat.)  Work distribution in the fit kernels is dynamic (waves pull tiles / voxels through
atomic counters): the per-voxel results must not depend on it.

    python scripts/fit_repeat_check.py [--reps 300]
"""
import argparse
import ctypes
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest(*arrs):
    h = hashlib.sha1()
    for a in arrs:
        h.update(np.ascontiguousarray(a).view(np.uint8).tobytes())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    import torch

    import bench
    from dosma_amd import _lib as L

    lib = L.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    st = torch.cuda.current_stream(dev)
    y = bench.make_volume(torch, dev, 20260928)
    n = y.shape[1]
    popt = torch.empty((n, 2), dtype=torch.float32, device=dev)
    r2 = torch.empty(n, dtype=torch.float32, device=dev)
    for recipe in ("A", "B"):
        classes = {}
        for rep in range(args.reps):
            a = bench.make_args(L, y, popt, r2, st.cuda_stream, recipe)
            a.device = 0
            L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
            torch.cuda.synchronize()
            key = (int(popt.view(torch.int32).to(torch.int64).sum().item()), int(r2.view(torch.int32).to(torch.int64).sum().item()))
            classes.setdefault(key, []).append(rep)
        print(f"monoexp_fit_device recipe {recipe}, 512x512x160x8: {args.reps} launches -> {len(classes)} distinct results", flush=True)
    rng = np.random.default_rng(1)
    # host entries on smaller inputs (uploads dominate): masked fit, bi-exponential, DESS, region statistics
    E, N = 8, 1 << 20
    x = np.arange(1, E + 1) * 10.0
    ys = (rng.uniform(300, 1500, N) * np.exp(-x[:, None] / rng.uniform(15, 80, N)) + 18 * rng.standard_normal((E, N))).astype(np.float32)
    ys[:, rng.random(N) < 0.3] = 0
    mask = rng.random(N) < 0.4
    reps = max(20, args.reps // 6)
    seen = set()
    for rep in range(reps):
        o = L.monoexp_fit_host(x, ys, mask=mask, init=L.INIT_LOGLIN, want_info=True)
        seen.add(digest(o["popt"], o["r2"], o["info"], o["nfev"]))
    print(f"monoexp_fit_host masked, polyfit init, {N} voxels: {reps} calls -> {len(seen)} distinct results", flush=True)
    xb = np.linspace(2.0, 120.0, 12)
    yb = (800 * np.exp(-xb[:, None] / rng.uniform(10, 25, N // 8)) + 500 * np.exp(-xb[:, None] / rng.uniform(50, 120, N // 8))
          + 5 * rng.standard_normal((12, N // 8)))
    seen = set()
    for rep in range(reps):
        o = L.lmfit_host("biexponential", xb, yb, [700.0, -1 / 15.0, 600.0, -1 / 80.0], want_info=True)
        seen.add(digest(o["popt"], o["r2"], o["info"], o["nfev"]))
    print(f"lmfit_host biexponential, {N // 8} voxels x 12 samples: {reps} calls -> {len(seen)} distinct results", flush=True)
    e1 = rng.uniform(100, 1000, (128, 128, 64)).astype(np.float32)
    e2 = (e1 * rng.uniform(0.2, 0.9, e1.shape)).astype(np.float32)
    seen = set()
    for rep in range(reps):
        seen.add(digest(L.dess_t2_host(e1, e2, 0.01, 1.2, 0.5, bounds=(0, 100), nan_to_num=0.0, decimals=3)))
    print(f"dess_t2_host 128x128x64: {reps} calls -> {len(seen)} distinct results", flush=True)
    vals = rng.uniform(0, 80, (256, 256, 80))
    labels = rng.integers(0, 5, vals.shape).astype(np.uint8)
    seen = set()
    for rep in range(reps):
        seen.add(digest(L.region_stats_host(vals, labels, keys=(1, 2, 3, 4), bounds=(0, 100))))
    print(f"region_stats_host 256x256x80, 4 labels: {reps} calls -> {len(seen)} distinct results", flush=True)


if __name__ == "__main__":
    main()
