"""PCIe-inclusive rate of the host-buffer entry (qmri_monoexp_fit_host): pageable numpy in, numpy out.
Reported in DESIGN.md §10; never bench.py's `value`."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib

def main():
    E, N = 8, 512 * 512 * 160
    rng = np.random.default_rng(0)
    x = np.arange(1, E + 1) * 10.0
    t2 = rng.uniform(20, 80, N).astype(np.float32)
    y = np.empty((E, N), np.float32)
    for e in range(E):
        y[e] = (1000.0 * np.exp(-x[e] / t2)).astype(np.float32) * (1 + 0.01 * rng.standard_normal(N, dtype=np.float32))
    post = dict(inv_abs_b=True, bounds=((-np.inf, np.inf), (0, 100.0)), r2_threshold=0.9, nan_to_num=0.0, decimals=1)
    for name, kw in (("A", {}), ("B", {"init": _lib.INIT_LOGLIN}),
                     ("B recipe (tc, r2 only)", {"init": _lib.INIT_LOGLIN, "post": post, "want_tc": True, "want_popt": False})):
        for rep in range(3):
            t0 = time.perf_counter()
            out = _lib.monoexp_fit_host(x, y, **kw)
            dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        _lib.monoexp_fit_host(x, y, out=out, **kw)
        dt2 = time.perf_counter() - t0
        print(f"run {name}: host entry {dt*1e3:.1f} ms  {N/dt:.3e} voxel-fits/s (fresh output arrays); "
              f"{dt2*1e3:.1f} ms {N/dt2:.3e}/s (outputs already touched)")

main()
