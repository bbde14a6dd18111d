"""Parity margins of the fit kernel on the golden sets (run on the GPU box): max / quantiles of the relative error of
popt against the reference's outputs, for A/B of numerics switches such as QMRI_FIT_UNIFORM_X."""
import numpy as np, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from dosma_amd import _lib as L
from conftest import rel_err
tag = os.environ.get("QMRI_FIT_UNIFORM_X")
g = np.load("tests/golden/g2_cfg2_8echo.npz")
for snr in (100, 50, 20):
    x, y = g["x"], g[f"y_snr{snr}"]
    for name, kw, ref in (("A", dict(p0=(1.0, -1 / 30.0)), g[f"popt_snr{snr}"]),):
        o = L.monoexp_fit_host(x, y, want_info=True, **kw)
        d = rel_err(o["popt"], ref).max(axis=1)
        d = d[np.isfinite(d)]
        print(f"uniform={tag} g2 snr{snr} {name}: max {d.max():.2e} q99.9 {np.quantile(d, 0.999):.2e} median {np.median(d):.2e} "
              f"nfev equal {(o['nfev'] == g[f'nfev_snr{snr}']).mean():.5f}")
g = np.load("tests/golden/g3_edges.npz")
x, y = g["x"], g["y"]
o = L.monoexp_fit_host(x, y, p0=(1.0, -1 / 30.0), want_info=True)
ier = g["ier"]; ok = (ier >= 1) & (ier <= 4)
same = ((o["info"] >= 1) & (o["info"] <= 4)) == ok
both = same & ok
d = rel_err(o["popt"][both], g["popt"][both]).max(axis=1)
print(f"uniform={tag} g3 pure-noise columns: n {both.sum()} same_class {same.mean():.4f} frac>1e-4 {(d > 1e-4).mean():.4f} frac>1e-3 {(d > 1e-3).mean():.4f}")
