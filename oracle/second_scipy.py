"""A second scipy build as a pin (TEST INFRASTRUCTURE; nothing under dosma_amd/ imports this).

The reference leaves scipy unpinned (/root/reference/requirements.txt:12, setup.py:108) and the image holds two builds:
1.15.3 (the C translation of MINPACK; what oracle/make_golden.py ran under) on /usr/bin/python3 and 1.7.1 (the FORTRAN
MINPACK -- what a DOSMA-0.1.2-era install ran) in /opt/conda.  This script is run BY THE OTHER INTERPRETER
(tests/test_oracle.py::test_fixtures_hold_under_the_fortran_minpack_scipy): it makes the reference's per-voxel call
(/root/reference/dosma/core/fitting.py:1030, `sop.curve_fit(func, x, y, p0=p0, maxfev=maxfev, ftol=ftol)`) -- through the
imported reference's own `curve_fit` when /root/reference is present, restated otherwise -- on the INPUTS of the committed
fixtures g2 (first `ncols` columns per SNR) and g3 (all columns), and writes popt / r2 / nfev / ier for the test to compare
with the fixtures' outputs.

    /opt/conda/bin/python3.9 oracle/second_scipy.py tests/golden out.npz [ncols]
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
P0 = (1.0, -1 / 30.0)
FTOL, MAXFEV, EPS = 1e-5, 100, 1e-8


def monoexponential(x, a, b):  # /root/reference/dosma/core/fitting.py:1081-1091
    return a * np.exp(b * x)


def per_voxel(func, x, y):
    """popt, r2 (the reference's rules, fitting.py:1026-1073) + scipy's nfev / ier for the same call."""
    from scipy import optimize as sop

    n = y.shape[1]
    popt, r2 = np.full((n, 2), np.nan), np.zeros(n)
    nfev, ier = np.zeros(n, np.int32), np.zeros(n, np.int32)
    for i in range(n):
        yi = y[:, i]
        if (yi == 0).all():
            continue
        try:
            out = sop.curve_fit(func, x, yi, p0=P0, maxfev=MAXFEV, ftol=FTOL, full_output=True)
        except RuntimeError:
            ier[i], nfev[i] = 5, -1
            continue
        popt[i], nfev[i], ier[i] = out[0], out[2]["nfev"], out[4]
        res = yi - func(x, *out[0])
        r2[i] = 1 - np.sum(res ** 2) / (np.sum((yi - np.mean(yi)) ** 2) + EPS)
    return popt, r2, nfev, ier


def main(golden_dir, out_path, ncols=4000):
    import scipy

    out = {"scipy_version": np.array(scipy.__version__), "via_reference": np.array(False)}
    ref_curve_fit = None
    func = monoexponential
    try:
        from oracle import ref_harness

        if ref_harness.reference_available():
            dosma = ref_harness.load_reference()
            ref_curve_fit, func = dosma.curve_fit, dosma.monoexponential
            out["via_reference"] = np.array(True)
    except Exception as e:  # the reference does not import under this interpreter: the restated call alone
        print("reference not importable here:", repr(e)[:200])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g2 = np.load(os.path.join(golden_dir, "g2_cfg2_8echo.npz"))
        g3 = np.load(os.path.join(golden_dir, "g3_edges.npz"))
        cases = [(f"g2_snr{snr}", g2["x"], g2[f"y_snr{snr}"][:, :ncols]) for snr in (100, 50, 20)]
        cases.append(("g3", g3["x"], g3["y"]))
        for tag, x, y in cases:
            popt, r2, nfev, ier = per_voxel(func, x, y)
            out[f"{tag}_popt"], out[f"{tag}_r2"], out[f"{tag}_nfev"], out[f"{tag}_ier"] = popt, r2, nfev, ier
            if ref_curve_fit is not None:  # the reference's own driver (fitting.py:752-870) under this scipy: same numbers
                rp, rr = ref_curve_fit(func, x, y, p0=P0)
                assert np.array_equal(rp, popt, equal_nan=True), tag
                assert np.allclose(rr, r2, rtol=0, atol=1e-6), tag  # (float32 ss_tot for float32 samples in the reference)
    np.savez(out_path, **out)
    print("wrote", out_path, "scipy", scipy.__version__, "via_reference", bool(out["via_reference"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4000)
