"""How far is the kernel from the reference on the pure-noise columns of golden g3, and is the difference the emulated forward
differences (C restatement, jac_mode = 2 = the kernel's arithmetic) or something else?  (tests/test_fit_gpu.py::test_edge_cases_golden)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from dosma_amd import _lib as L
import oracle.fit_oracle as fo
from conftest import rel_err

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g3_edges.npz"))
x, y = g["x"], g["y"]
P0 = (1.0, -1 / 30.0)
o = L.monoexp_fit_host(x, y, p0=P0, want_info=True)
ok_ref = (g["ier"] >= 1) & (g["ier"] <= 4)
ok = (o["info"] >= 1) & (o["info"] <= 4)
both = ok & ok_ref
d = rel_err(o["popt"][both], g["popt"][both]).max(axis=1)
print(f"vs the reference (scipy): {both.sum()} columns, same class {np.mean(ok == ok_ref):.4f}, > 1e-4: {np.mean(d > 1e-4):.4f}, > 1e-3: {np.mean(d > 1e-3):.4f}")
popt, r2, info, nfev = fo.curve_fit_c(x, y, P0, jac_mode=2, full_output=True)
okc = (info >= 1) & (info <= 4)
b2 = ok & okc
d2 = rel_err(o["popt"][b2], popt[b2]).max(axis=1)
print(f"vs the C restatement with the kernel's emulated differences: same class {np.mean(ok == okc):.4f}, nfev equal {np.mean(o['nfev'][b2] == nfev[b2]):.4f}, > 1e-8: {np.mean(d2 > 1e-8):.4f}, > 1e-4: {np.mean(d2 > 1e-4):.4f}, > 1e-3: {np.mean(d2 > 1e-3):.4f}")
popt0, r20, info0, nfev0 = fo.curve_fit_c(x, y, P0, jac_mode=0, full_output=True)
ok0 = (info0 >= 1) & (info0 <= 4)
b3 = ok_ref & ok0
d3 = rel_err(popt0[b3], g["popt"][b3]).max(axis=1)
print(f"C restatement with TRUE differences vs the reference: > 1e-4: {np.mean(d3 > 1e-4):.4f}, > 1e-3: {np.mean(d3 > 1e-3):.4f}")
