import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from dosma_amd import _lib as L
from oracle import fit_oracle as fo
from conftest import rel_err
rng = np.random.default_rng(11)
E, N = 8, 4000
x = np.arange(1, E + 1) * 10.0
y = rng.uniform(300, 1500, N) * np.exp(-x[:, None] / rng.uniform(15, 80, N)) + 8 * rng.standard_normal((E, N))
for scale in (1e-30, 1e-20, 1e-16):
    ys = y * scale
    o = L.monoexp_fit_host(x, ys, p0=(1.0, -1 / 30.0), want_info=True)
    popt, r2, info, nfev = fo.curve_fit_c(x, ys, (1.0, -1 / 30.0), jac_mode=2, full_output=True)
    d = rel_err(o["popt"], popt).max(axis=1)
    print(os.environ.get("QMRI_FIT_LMPAR_CF"), os.environ.get("QMRI_FIT_UNIFORM_X"), "scale", scale, "frac>1e-4", (d > 1e-4).mean(),
          "gpu", o["popt"][0], o["info"][0], o["nfev"][0], "oracle", popt[0], info[0], nfev[0])
