"""Time the general lmdif kernel (lm_generic.hip) on device-resident data: bi-exponential 12-echo slab and the
true-forward-difference mono-exponential on the bench's 8-echo volume (next to the fast kernel)."""
import argparse, ctypes, hashlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 22)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
lib = L.load()
dev = torch.device("cuda", 0)


def run(model, x, y, p0):
    n = y.shape[1]
    npar = L.MODEL_NPARAMS[model]
    popt = torch.empty((n, npar), dtype=torch.float64, device=dev)
    r2 = torch.empty(n, dtype=torch.float64, device=dev)
    nfev = torch.empty(n, dtype=torch.int16, device=dev)
    a = L.QmriLmfitArgs()
    lib.qmri_lmfit_defaults(ctypes.byref(a))
    a.model = L.MODELS[model]
    a.y, a.y_dtype, a.E, a.N, a.ld = y.data_ptr(), L.QMRI_F32, y.shape[0], n, n
    xs = np.ascontiguousarray(x, np.float64)
    a.x = xs.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    for j, v in enumerate(p0):
        a.p0[j] = v
    a.popt, a.r2, a.nfev = popt.data_ptr(), r2.data_ptr(), nfev.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    ts = []
    for _ in range(args.reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        L.check(lib.qmri_lmfit_device(ctypes.byref(a), None))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    fitted = (nfev > 0).sum().item()
    print(f"{model:16s} E={y.shape[0]:2d} n={n} best {min(ts)*1e3:8.2f} ms  {n/min(ts)/1e6:8.1f} Mvox/s  "
          f"fitted {fitted/n:.2f}  mean nfev {nfev[nfev > 0].float().mean().item():.1f}  "
          f"nan {torch.isnan(popt[:, 0]).float().mean().item():.3f}  "
          f"sha1 {hashlib.sha1(popt.cpu().numpy().tobytes() + r2.cpu().numpy().tobytes() + nfev.cpu().numpy().tobytes()).hexdigest()[:16]}")


# bi-exponential: two compartments, 12 echoes
g = torch.Generator(device=dev); g.manual_seed(7)
n = args.n
x = np.linspace(4.0, 92.0, 12)
xt = torch.tensor(x, device=dev, dtype=torch.float32)[:, None]
a1 = torch.empty(n, device=dev).uniform_(300, 900, generator=g)
a2 = torch.empty(n, device=dev).uniform_(200, 700, generator=g)
ts_ = torch.empty(n, device=dev).uniform_(5, 15, generator=g)
tl = torch.empty(n, device=dev).uniform_(40, 90, generator=g)
y = a1 * torch.exp(-xt / ts_) + a2 * torch.exp(-xt / tl) + 2.0 * torch.randn((12, n), device=dev, generator=g)
y[:, : n // 4] = 0  # background
run("biexponential", x, y.contiguous(), (500.0, -0.1, 500.0, -0.02))

y8 = bench.make_volume(torch, dev, 20260928)[:, :n].contiguous()
run("monoexponential", bench.TE, y8, (1.0, -1 / 30.0))
