"""Fit rate per echo count (device-resident fp32 volume of 2^23 voxels, recipe A): exercises every EMAX variant."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dosma_amd import _lib as L
import bench
lib = L.load()
dev = torch.device("cuda", 0)
n = 1 << 23
for E in (4, 8, 12, 16, 20, 32):
    g = torch.Generator(device=dev); g.manual_seed(E)
    x = np.arange(1, E + 1) * (80.0 / E)
    s0 = torch.rand(n, device=dev, generator=g) * 1200 + 300
    t2 = torch.rand(n, device=dev, generator=g) * 65 + 15
    xs = torch.tensor(x, device=dev, dtype=torch.float32)[:, None]
    y = (s0 * torch.exp(-xs / t2) + 18 * torch.randn((E, n), device=dev, generator=g)).float().contiguous()
    popt = torch.empty((n, 2), dtype=torch.float32, device=dev); r2 = torch.empty(n, dtype=torch.float32, device=dev)
    a = bench.make_args(L, y, popt, r2, torch.cuda.current_stream().cuda_stream, "A")
    xa = (ctypes.c_double * E)(*x); a.x = xa; a.E = E
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"E={E:2d} {lib.qmri_monoexp_kernel_name(ctypes.byref(a)).decode():28s} {min(ts)*1e3:7.2f} ms  {n/min(ts)/1e6:8.1f} Mvox/s")
