"""Bit-level fingerprint + timing of the fit kernel under the library DOSMA_AMD_LIB points at (same-box A/B of two builds:
run once per library and compare the SHA-1 columns).  RAW outputs (no post-processing, float64 (a, b), r2, stop code,
evaluation count): the full bench volume with the fixed and the log-linear start, a non-uniform 4-echo slab (one exp per
sample) and a 12-echo slab (the two-waves-per-SIMD instantiation)."""
import ctypes, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dosma_amd import _lib as L
import bench

lib = L.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream


def run(tag, y, x, init, reps=4):
    e, n = y.shape
    popt = torch.empty((n, 2), dtype=torch.float64, device=dev)
    r2 = torch.empty(n, dtype=torch.float64, device=dev)
    info = torch.zeros(n, dtype=torch.uint8, device=dev)
    nfev = torch.zeros(n, dtype=torch.int16, device=dev)
    a = L.default_args()
    xs = np.ascontiguousarray(x, dtype=np.float64)
    a.y, a.y_dtype, a.E, a.N, a.ld = y.data_ptr(), L.QMRI_F32, e, n, n
    a.x = xs.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    a.popt, a.r2, a.out_dtype = popt.data_ptr(), r2.data_ptr(), L.QMRI_F64
    a.info, a.nfev = info.data_ptr(), nfev.data_ptr()
    a.stream = st
    if init == "scalar":
        a.init, a.a0, a.b0 = L.INIT_SCALAR, bench.P0_A[0], bench.P0_A[1]
    else:
        a.init = L.INIT_LOGLIN
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    h = hashlib.sha1()
    for t in (popt, r2, info, nfev):
        h.update(t.cpu().numpy().tobytes())
    print(f"{tag:34s} {min(ts) * 1e3:8.3f} ms   sha1 {h.hexdigest()[:16]}   mean nfev {nfev.double().mean().item():.3f}   "
          f"{lib.qmri_monoexp_kernel_name(ctypes.byref(a)).decode()}")


y = bench.make_volume(torch, dev, 20260928)
run("8 echoes uniform, fixed p0", y, bench.TE, "scalar")
run("8 echoes uniform, log-linear p0", y, bench.TE, "loglin")
n4 = 1 << 22
run("4 echoes non-uniform (1,10,30,60)", y[[0, 1, 3, 6], :n4].contiguous(), [1.0, 10.0, 30.0, 60.0], "loglin")
g = torch.Generator(device=dev).manual_seed(7)
x12 = np.arange(1, 13) * 7.5
t2 = torch.empty(n4, device=dev).uniform_(15, 80, generator=g)
s0 = torch.empty(n4, device=dev).uniform_(300, 1500, generator=g)
y12 = (s0[None] * torch.exp(-torch.tensor(x12, device=dev, dtype=torch.float32)[:, None] / t2[None])
       + torch.randn(12, n4, device=dev, generator=g) * 10).float().contiguous()
run("12 echoes uniform, fixed p0", y12, x12, "scalar")
x12n = x12.copy(); x12n[5] += 1.0
run("12 echoes non-uniform, fixed p0", y12, x12n, "scalar")
