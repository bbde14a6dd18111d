import sys, ctypes; sys.path.insert(0,'.')
import numpy as np, torch
from dosma_amd import _lib as L
from oracle import fit_oracle as fo
N, E = 512 * 512 * 160, 8
x = np.arange(1, 9) * 10.0
gen = torch.Generator(device="cuda").manual_seed(20260928)
s0 = torch.rand(N, device="cuda", generator=gen, dtype=torch.float64) * 1200 + 300
t2 = torch.rand(N, device="cuda", generator=gen, dtype=torch.float64) * 65 + 15
bg = torch.rand(N, device="cuda", generator=gen) < 0.3
xt = torch.tensor(x, device="cuda", dtype=torch.float64)
y = (s0[None, :] * torch.exp(-xt[:, None] / t2[None, :]))
y[:, bg] = 0
y = y.to(torch.float32).contiguous()
popt = torch.empty((N, 2), dtype=torch.float64, device="cuda"); r2 = torch.empty(N, dtype=torch.float64, device="cuda")
info = torch.empty(N, dtype=torch.int8, device="cuda"); nfev = torch.empty(N, dtype=torch.int16, device="cuda")
a = L.default_args()
a.y, a.y_dtype, a.E, a.N, a.ld = y.data_ptr(), L.QMRI_F32, E, N, N
a.x = x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); a.a0, a.b0 = 1.0, -1/30
a.popt, a.r2, a.out_dtype = popt.data_ptr(), r2.data_ptr(), L.QMRI_F64
a.info, a.nfev = info.data_ptr(), nfev.data_ptr()
L.check(L.load().qmri_monoexp_fit_device(ctypes.byref(a), None)); torch.cuda.synchronize()
fg = ~bg
tc = 1 / popt[:, 1].abs()
err = ((tc - t2).abs() / t2)
bad = torch.nonzero(fg & ~(err < 1e-3)).flatten()
print("bad", bad.numel(), "info hist", torch.bincount(info.long() + 1).tolist())
for b in bad[:10].tolist():
    yy = y[:, b:b+1].cpu().numpy()
    r = [fo.curve_fit_c(x, yy, (1.0, -1/30), jac_mode=m, full_output=True) for m in (0, 2)]
    print(b, s0[b].item(), t2[b].item(), yy.ravel()[:3], "gpu", popt[b].tolist(), info[b].item(), nfev[b].item(), "| fd", r[0][0][0], r[0][2][0], r[0][3][0], "| emu", r[1][0][0], r[1][2][0], r[1][3][0])
import os
os.makedirs('gpurun_out', exist_ok=True)
np.save('gpurun_out/bad_y.npy', y[:, bad].cpu().numpy())
yy = y[:, bad[:1]].cpu().numpy()
for rep in (1, 64, 257):
    o = L.monoexp_fit_host(x, np.repeat(yy, rep, axis=1), p0=(1.0, -1/30), want_info=True)
    print("isolated x", rep, o['popt'][0], o['info'][0], o['nfev'][0], "unique", np.unique(o['nfev']))
