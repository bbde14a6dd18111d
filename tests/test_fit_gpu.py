"""GPU parity tests: HIP kernel (through the C ABI) vs the oracle and the golden vectors produced by
the real reference.  Tolerance is BASELINE.json's: popt / r2 within 1e-4 relative of
scipy.optimize.curve_fit (floating-point path -> not bit-exact; integer outputs -- info, nfev -- are
compared exactly where the decisions are not borderline).

Run on the GPU box with ``pytest -m gpu``.  Nothing here reads /root/reference.
"""
import ctypes
import os

import numpy as np
import pytest

from dosma_amd import _lib as L
from oracle import fit_oracle as fo

from _sweeps import SHAPES_AND_DTYPES, shape_sweep_case

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star: "popt/r2 match scipy.optimize.curve_fit within 1e-4 rel"
P0 = (1.0, -1 / 30.0)
POST_A = dict(inv_abs_b=True, bounds=((-np.inf, np.inf), (0, 100.0)), r2_threshold=0.9,
              nan_to_num=0.0, decimals=1)


def post(bounds=(0, 100.0), dp=3, thr=0.9):
    return dict(inv_abs_b=True, bounds=((-np.inf, np.inf), bounds), r2_threshold=thr,
                nan_to_num=0.0, decimals=dp)


def r2_close(a, b):
    # r2 = 1 - ss_res/(ss_tot+eps) is O(1) near 1; 1e-4 relative on r2 ~ 1e-4 absolute.  For wildly
    # negative r2 (bad fits) the relative form is the meaningful one.
    return np.abs(a - b) <= RTOL * np.maximum(1.0, np.abs(b))


# ------------------------------------------------------------------------------- golden: headline
@pytest.mark.parametrize("snr", [100, 50, 20])
def test_raw_fit_vs_reference_golden(golden, relerr, snr, test_device):
    g = golden("g2_cfg2_8echo.npz")
    x, y = g["x"], g[f"y_snr{snr}"]
    o = L.monoexp_fit_host(x, y, p0=P0, want_info=True, device=test_device)  # (QMRI_TEST_DEVICE: a non-zero ordinal on a multi-GPU box)
    d = relerr(o["popt"], g[f"popt_snr{snr}"]).max(axis=1)
    assert d.max() < RTOL, f"max rel {d.max()}"
    assert r2_close(o["r2"], g[f"r2_snr{snr}"]).all()
    # same MINPACK decisions: success/failure class identical, nfev identical
    ier = g[f"ier_snr{snr}"]
    assert (((o["info"] >= 1) & (o["info"] <= 4)) == ((ier >= 1) & (ier <= 4))).all()
    assert (o["nfev"] == g[f"nfev_snr{snr}"]).mean() > 0.999
    assert (o["info"][ier == 0] == 0).all()  # skipped background voxels


@pytest.mark.parametrize("snr", [100, 50, 20])
def test_recipes_vs_reference_golden(golden, snr):
    """MonoExponentialFit defaults (A) and tc0='polyfit', dp=3 (B): fused init + fit + post-process."""
    g = golden("g2_cfg2_8echo.npz")
    x, y = g["x"], g[f"y_snr{snr}"]
    a = L.monoexp_fit_host(x, y, p0=P0, post=POST_A, want_tc=True)
    b = L.monoexp_fit_host(x, y, init=L.INIT_LOGLIN, post=post(), want_tc=True)
    # rounded maps: equal except where tc sits within 1e-4 rel of a rounding boundary
    assert np.abs(a["tc"] - g[f"tcA_snr{snr}"]).max() <= 0.1 + 1e-9
    assert (a["tc"] != g[f"tcA_snr{snr}"]).mean() < 2e-3
    assert np.abs(b["tc"] - g[f"tcB_snr{snr}"]).max() <= 1e-3 + 1e-9
    assert (b["tc"] != g[f"tcB_snr{snr}"]).mean() < 2e-3
    assert r2_close(a["r2"], g[f"r2A_snr{snr}"]).all()
    assert r2_close(b["r2"], g[f"r2B_snr{snr}"]).all()
    # background (all-zero) voxels: tc 0, r2 0 (nan_to_num after the skip rule)
    bg = (y == 0).all(axis=0)
    assert (a["tc"][bg] == 0).all() and (a["r2"][bg] == 0).all()


def test_tests_generator_golden(golden, relerr):
    """G1 = the reference tests' generator (tests/core/test_fitting.py:18-31, 199-277), seeded."""
    g = golden("g1_tests_generator.npz")
    x, y = g["x"], g["y"].reshape(4, -1)
    t = (1 / np.abs(g["b"])).reshape(-1)
    o = L.monoexp_fit_host(x, y, p0=P0, post=post(dp=8), want_tc=True)
    assert np.allclose(o["tc"], t)
    assert relerr(o["tc"], g["tc_default"].reshape(-1)).max() < RTOL
    o = L.monoexp_fit_host(x, y, init=L.INIT_LOGLIN, post=post(dp=8), want_tc=True)
    assert relerr(o["tc"], g["tc_polyfit"].reshape(-1)).max() < RTOL
    o = L.monoexp_fit_host(x, y, p0=(1.0, 1.0))
    assert relerr(o["popt"], g["popt"].reshape(-1, 2)).max() < RTOL
    assert r2_close(o["r2"], g["r2"].reshape(-1)).all()
    # mask: fit only selected voxels, fill = nan_to_num (0.0) elsewhere -- for tc AND r2
    m = g["mask"].reshape(-1)
    o = L.monoexp_fit_host(x, y, p0=P0, mask=m, post=post(dp=8), want_tc=True, want_info=True)
    assert relerr(o["tc"], g["tc_masked"].reshape(-1)).max() < RTOL
    assert (o["tc"][~m] == 0).all() and (o["r2"][~m] == 0).all() and (o["info"][~m] == -1).all()
    # raw fit with a mask: NaN outside (reference CurveFitter, nan_to_num=None)
    o = L.monoexp_fit_host(x, y, p0=(1.0, 1.0), mask=m)
    assert np.isnan(o["popt"][~m]).all() and np.isnan(o["r2"][~m]).all()
    assert relerr(o["popt"][m], g["popt_masked"].reshape(-1, 2)[m]).max() < RTOL
    # zeros in echo 0 (test_fitting.py:267-277)
    o = L.monoexp_fit_host(x, g["y_zero_echo0"].reshape(4, -1), init=L.INIT_LOGLIN, post=post(dp=8),
                           want_tc=True)
    assert relerr(o["tc"], g["tc_zero_echo0"].reshape(-1)).max() < RTOL


def test_edge_cases_golden(golden, relerr):
    g = golden("g3_edges.npz")
    x, y = g["x"], g["y"]
    o = L.monoexp_fit_host(x, y, p0=P0, want_info=True)
    ier = g["ier"]
    ok = (ier >= 1) & (ier <= 4)
    assert (((o["info"] >= 1) & (o["info"] <= 4)) == ok)[:8].all()
    assert o["info"][0] == 0 and np.isnan(o["popt"][0]).all() and o["r2"][0] == 0  # skip rule
    assert o["info"][5] == 5 and np.isnan(o["popt"][5]).all() and o["r2"][5] == 0  # maxfev -> NaN, 0
    assert np.allclose(o["popt"][:8], g["popt"][:8], rtol=RTOL, atol=1e-8, equal_nan=True)
    # pure-noise columns: not identifiable, decisions are chaotic at the 1e-8 level in BOTH solvers;
    # require the same success pattern and agreement on the overwhelming majority.  (The few-percent tail
    # moves with ANY reordering of the arithmetic -- 12, 16, 18 of these 473 columns with libm exponentials +
    # qrsolv, the power chain, the closed-form lmpar iteration -- while on tissue-like data (g2) the same three
    # variants agree with the reference to 2e-6 at worst with identical nfev: scripts/edge_frac.py.)
    same_class = ((o["info"] >= 1) & (o["info"] <= 4)) == ok
    assert same_class.mean() > 0.99
    both = same_class & ok
    d = relerr(o["popt"][both], g["popt"][both]).max(axis=1)
    # (how loose is 5 %?  Two builds of scipy itself -- 1.15.3 and the Fortran-MINPACK 1.7.1 -- differ beyond 1e-4 on 4.2 % of these same
    #  columns with identical stop codes: tests/test_oracle.py::test_fixtures_hold_under_the_fortran_minpack_scipy)
    assert (d > RTOL).mean() < 0.05          # measured 3.8 % (18 of 473; scripts/edge_noise_check.py)
    assert (d > 10 * RTOL).mean() < 0.02     # measured 1.3 %
    # ... and the tail is the ill-conditioning of fitting noise, not a different algorithm: against the C restatement of lmdif run
    # with the kernel's own difference quotients (jac_mode 2) every column takes the SAME number of evaluations and stops with the
    # same code -- the decision path is identical, only last-digit arithmetic (the exponential) differs, amplified on flat minima;
    # the restatement with TRUE differences is itself 0.85 % / 0.63 % away from scipy on these columns
    popt_c, _, info_c, nfev_c = fo.curve_fit_c(x, y, P0, jac_mode=2, full_output=True)
    ok_c = (info_c >= 1) & (info_c <= 4)
    assert ((((o["info"] >= 1) & (o["info"] <= 4)) == ok_c).mean() > 0.995)
    b2 = ok_c & (o["info"] >= 1) & (o["info"] <= 4)
    assert (o["nfev"][b2] == nfev_c[b2]).mean() > 0.995
    # p0 = None (ones) and the reference tests' far guess p0 = (1, 50) with x = 1..4
    o = L.monoexp_fit_host(x, y, p0=(1.0, 1.0))
    both = ~np.isnan(g["popt_p0none"][:, 0]) & ~np.isnan(o["popt"][:, 0])
    assert (np.isnan(g["popt_p0none"][:, 0]) == np.isnan(o["popt"][:, 0])).mean() > 0.99
    o = L.monoexp_fit_host(g["x4"].astype(float), g["y4"], p0=(1.0, 50.0))
    assert (np.isnan(o["popt"]) == np.isnan(g["popt_1_50"])).all()
    o = L.monoexp_fit_host(g["x4"].astype(float), g["y4"], p0=(1.0, 1.0))
    assert relerr(o["popt"], g["popt_1_1"]).max() < RTOL
    # int16 samples (DICOM-like)
    o = L.monoexp_fit_host(x, g["y_int16"], p0=P0)
    assert relerr(o["popt"], g["popt_int16"]).max() < RTOL
    assert r2_close(o["r2"], g["r2_int16"]).all()
    o = L.monoexp_fit_host(x, g["y_int16"], init=L.INIT_LOGLIN, post=post(), want_tc=True)
    assert (o["tc"] != g["tc_int16"]).mean() < 5e-3
    # Cones recipe: ub = inf lets inf through to nan_to_num -> DBL_MAX
    o = L.monoexp_fit_host(x, y, init=L.INIT_LOGLIN, post=post(bounds=(0, np.inf)), want_tc=True)
    assert (o["tc"] != g["tc_cones"]).mean() < 0.02


def test_scan_recipes_golden(golden, relerr):
    g = golden("g4_recipes.npz")
    o = L.monoexp_fit_host(g["tsl"], g["y"].reshape(4, -1), mask=g["mask"].reshape(-1),
                           init=L.INIT_LOGLIN, post=post(bounds=(0, 500)), want_tc=True)
    assert (o["tc"] != g["tc"].reshape(-1)).mean() < 2e-3
    assert np.abs(o["tc"] - g["tc"].reshape(-1)).max() <= 1e-3 + 1e-9
    assert r2_close(o["r2"], g["r2"].reshape(-1)).all()
    o = L.monoexp_fit_host(g["te_mapss"], g["y_mapss"].reshape(4, -1), init=L.INIT_LOGLIN,
                           post=post(bounds=(0, 100)), want_tc=True)
    assert (o["tc"] != g["tc_mapss"].reshape(-1)).mean() < 2e-3


# ------------------------------------------------------------------------------- vs the oracle
def vs_true_lmdif(relerr, o, x, y, p0, frac=1e-3, cls=0.999):
    """the sweeps below compare the kernel with the oracle's mode 2 (lmdif with the kernel's own difference
    quotients: the comparator that shows the DECISION path is the same, nfev for nfev).  Mode 2 is written to mimic the kernel, so
    every sweep also holds the 1e-4 bar against mode 0 -- MINPACK's true forward differences, i.e. what scipy / the reference run,
    pinned to g2 / g3 and two scipy builds in tests/test_oracle.py -- on the same columns."""
    popt, r2, info, nfev = fo.curve_fit_c(x, y, p0, jac_mode=0, full_output=True)
    ok_k, ok_o = (o["info"] >= 1) & (o["info"] <= 4), (info >= 1) & (info <= 4)
    assert (ok_k == ok_o).mean() > cls, (ok_k == ok_o).mean()
    both = ok_k & ok_o
    d = relerr(o["popt"][both], popt[both]).max(axis=1)
    assert (d > RTOL).mean() < frac, f"vs true lmdif: frac>{RTOL}: {(d > RTOL).mean()}, max {d.max()}"
    assert r2_close(o["r2"][both], r2[both]).mean() > cls
    return popt, r2, info, nfev


# ------------------------------------------------------------------------------- vs the oracle
@pytest.mark.parametrize("E,dtype", SHAPES_AND_DTYPES)
def test_vs_oracle_shapes_and_dtypes(relerr, E, dtype):
    """Every kernel variant (EMAX 4/8/16/32, full/partial, f32/f64 staging) and input dtype."""
    x, y = shape_sweep_case(E, dtype)
    o = L.monoexp_fit_host(x, y, p0=P0, want_info=True)
    popt, r2, info, nfev = fo.curve_fit_c(x, y, P0, jac_mode=2, full_output=True)
    same = ((o["info"] >= 1) & (o["info"] <= 4)) == ((info >= 1) & (info <= 4))
    assert same.mean() > 0.999
    d = relerr(o["popt"][same], popt[same]).max(axis=1)
    assert (d > RTOL).mean() < 1e-3, f"E={E}: frac>{RTOL}: {(d > RTOL).mean()}, max {d.max()}"
    assert r2_close(o["r2"][same], r2[same]).mean() > 0.999
    assert (o["nfev"] == nfev).mean() > 0.995
    vs_true_lmdif(relerr, o, x, y, P0)
    b = L.monoexp_fit_host(x, y, init=L.INIT_LOGLIN, post=post(), want_tc=True)
    tc, r2o, _ = fo.monoexp_fit_arrays(x, y, tc0="polyfit", decimal_precision=3, jac_mode=2)
    assert (np.abs(b["tc"] - tc) > 1e-3 + 1e-9).mean() < 1e-3
    tc0, _, _ = fo.monoexp_fit_arrays(x, y, tc0="polyfit", decimal_precision=3, jac_mode=0)
    assert (np.abs(b["tc"] - tc0) > 1e-3 + 1e-9).mean() < 1e-3


@pytest.mark.parametrize("x", [
    np.arange(1, 9) * 10.0,              # x_0 = 1 * step: one exponential per evaluation (FitKArgs::x0_pow = 1)
    np.arange(0, 8) * 10.0,              # x_0 = 0: e_0 = 1
    np.arange(3, 9) * 4.9,               # x_0 = 3 * step, step not exactly representable (8-ulp rule), E = 6
    5.0 + np.arange(8) * 10.0,           # equally spaced, x_0 not a multiple of the step: two exponentials
    7.0 + np.arange(5) * 9.0,            # the same with the partial-E kernel variant
    np.arange(1, 17) * 5.0,              # E = 16: the depth-5 product tree
    np.array([10., 20., 30., 40., 50., 60., 70., 80.000001]),  # NOT equally spaced by 1e-8: E exponentials
    -10.0 + np.arange(8) * 10.0,         # x_0 < 0: the chain is off (0 * inf hazard), E exponentials
    np.array([40., 10., 80., 20., 60., 30., 70., 50.]),         # not sorted (Mapss sorts, curve_fit callers need not)
    np.array([10., 10., 20., 20., 40., 40., 80., 80.]),         # repeated sample times
    np.arange(1, 9) * 1e4,               # times in microseconds: b ~ 1e-5
], ids=["k1", "k0", "k3", "offset", "offset5", "e16", "almost", "negative", "unsorted", "repeated", "microseconds"])
def test_equally_spaced_sample_times_vs_oracle(relerr, x):
    """The exponential power chain of the fit kernel (equally spaced x: exp(b x_i) = exp(b x_0) exp(b step)^i) against
    the oracle, which always evaluates E exponentials: same solutions to the parity bar, same nfev."""
    rng = np.random.default_rng(len(x))
    E, N = len(x), 6000
    xe = (np.abs(x) + 1.0) * (80.0 / np.abs(x).max())  # generate with a decay over the sampled range whatever its unit
    y = rng.uniform(300, 1500, N) * np.exp(-xe[:, None] / rng.uniform(15, 80, N)) + 8 * rng.standard_normal((E, N))
    y = y.astype(np.float32)
    y[:, ::13] = 0
    o = L.monoexp_fit_host(x, y, p0=P0, want_info=True)
    popt, r2, info, nfev = fo.curve_fit_c(x, y, P0, jac_mode=2, full_output=True)
    same = ((o["info"] >= 1) & (o["info"] <= 4)) == ((info >= 1) & (info <= 4))
    assert same.mean() > 0.999
    d = relerr(o["popt"][same], popt[same]).max(axis=1)
    assert (d > RTOL).mean() < 1e-3, f"frac>{RTOL}: {(d > RTOL).mean()}, max {d.max()}"
    assert r2_close(o["r2"][same], r2[same]).mean() > 0.999
    assert (o["nfev"] == nfev).mean() > 0.995
    vs_true_lmdif(relerr, o, x, y, P0)


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1e12, 1e30])
def test_extreme_magnitudes_vs_oracle(relerr, scale):
    """Samples far from O(1): the guarded fast paths of the kernel (closed-form lmpar inside |.| < 1e100, reciprocals /
    square roots inside 1e-290..1e290, the exponential chain) must hand over to their slow forms where needed and agree
    with the oracle, which has none of them.  float64 samples; the amplitude scales, the rate does not."""
    rng = np.random.default_rng(11)
    E, N = 8, 4000
    x = np.arange(1, E + 1) * 10.0
    y = rng.uniform(300, 1500, N) * np.exp(-x[:, None] / rng.uniform(15, 80, N)) + 8 * rng.standard_normal((E, N))
    y[:, ::9] = 0
    y[:, 5::97] *= -1  # a few negative columns (fail or give nonsense in both solvers alike)
    ys = y * scale
    o = L.monoexp_fit_host(x, ys, p0=P0, want_info=True)
    popt, r2, info, nfev = fo.curve_fit_c(x, ys, P0, jac_mode=2, full_output=True)
    same = ((o["info"] >= 1) & (o["info"] <= 4)) == ((info >= 1) & (info <= 4))
    assert same.mean() > 0.995, same.mean()
    ok = same & (info >= 1) & (info <= 4)
    if scale >= 1e-16:
        d = relerr(o["popt"][ok], popt[ok]).max(axis=1)
        assert (d > RTOL).mean() < 5e-3, f"scale {scale}: frac>{RTOL}: {(d > RTOL).mean()}, max {d.max()}"
        vs_true_lmdif(relerr, o, x, ys, P0, frac=5e-3, cls=0.995)
    else:
        # samples ~1e-27 against p0 = (1, -1/30): the first step takes a to ~0 (1e-17: the rounding residue of 1 - 1),
        # where dF/db = a x e vanishes and b is no longer identifiable; lmdif stops on xtol after 7-13 evaluations
        # with b still at its initial value and a = O(1e-17) noise -- in the reference as here (scripts/tiny_probe.py).
        # What the map shows (tc = 1/|b|, r2 ~ 1) agrees; the noise in a does not and cannot.
        assert relerr(o["popt"][ok][:, 1], popt[ok][:, 1]).max() < RTOL
        assert np.abs(o["popt"][ok][:, 0]).max() < 1e-15 and np.abs(popt[ok][:, 0]).max() < 1e-15
    assert r2_close(o["r2"][ok], r2[ok]).mean() > 0.995
    assert np.isnan(o["popt"][~((o["info"] >= 1) & (o["info"] <= 4))]).all()
    b = L.monoexp_fit_host(x, ys, init=L.INIT_LOGLIN, post=post(), want_tc=True)
    tc, _, _ = fo.monoexp_fit_arrays(x, ys, tc0="polyfit", decimal_precision=3, jac_mode=2)
    assert (np.abs(b["tc"] - tc) > 1e-3 + 1e-9).mean() < 5e-3


def test_per_voxel_p0_and_y_bounds(relerr):
    rng = np.random.default_rng(5)
    N, E = 3000, 6
    x = np.linspace(5, 60, E)
    y = rng.uniform(300, 1500, N) * np.exp(-x[:, None] / rng.uniform(15, 80, N))
    y += 5 * rng.standard_normal((E, N))
    a0 = rng.uniform(200, 1600, N)
    b0 = -1 / rng.uniform(10, 90, N)
    o = L.monoexp_fit_host(x, y, init=L.INIT_PER_VOXEL, a0v=a0, b0v=b0)
    popt, r2 = fo.curve_fit_c(x, y, (a0, b0), jac_mode=2)
    assert relerr(o["popt"], popt).max() < RTOL
    popt0, _ = fo.curve_fit_c(x, y, (a0, b0), jac_mode=0)   # true lmdif (what scipy runs): the same bar
    assert relerr(o["popt"], popt0).max() < RTOL
    o = L.monoexp_fit_host(x, y, init=L.INIT_PER_VOXEL, p0=(1.0, 1.0), b0v=b0)  # mix scalar + array
    popt, r2 = fo.curve_fit_c(x, y, (1.0, b0), jac_mode=2)
    ok = ~np.isnan(popt[:, 0]) & ~np.isnan(o["popt"][:, 0])
    assert ok.mean() > 0.95 and relerr(o["popt"][ok], popt[ok]).max() < RTOL
    popt0, _ = fo.curve_fit_c(x, y, (1.0, b0), jac_mode=0)
    ok0 = ok & ~np.isnan(popt0[:, 0])
    assert ok0.mean() > 0.95 and (relerr(o["popt"][ok0], popt0[ok0]).max(axis=1) > RTOL).mean() < 1e-3
    # y_bounds: a voxel with any sample outside is skipped like an all-zero one
    o = L.monoexp_fit_host(x, y, p0=P0, y_bounds=(0.0, 1200.0), want_info=True)
    oob = ((y < 0) | (y > 1200)).any(axis=0)
    assert oob.any() and (~oob).any()
    assert np.isnan(o["popt"][oob]).all() and (o["r2"][oob] == 0).all() and (o["info"][oob] == 0).all()
    assert (~np.isnan(o["popt"][~oob, 0])).mean() > 0.99


def test_nonfinite_input_raises_like_reference():
    x = np.arange(1, 5) * 10.0
    y = np.ones((4, 1000), dtype=np.float32) * 100
    y[2, 77] = np.nan
    with pytest.raises(ValueError):
        L.monoexp_fit_host(x, y, p0=P0)
    y[2, 77] = np.inf
    with pytest.raises(ValueError):
        L.monoexp_fit_host(x, y, p0=P0)
    # ... but a non-finite sample OUTSIDE the mask never reaches the solver in the reference either
    m = np.ones(1000, dtype=bool)
    m[77] = False
    L.monoexp_fit_host(x, y, p0=P0, mask=m)


def test_argument_errors():
    x = np.arange(1, 5) * 10.0
    y = np.ones((4, 10), dtype=np.float32)
    with pytest.raises(ValueError):
        L.monoexp_fit_host(x[:3], y)  # len(x) != E
    with pytest.raises(ValueError):
        L.monoexp_fit_host(x[:1], y[:1])  # fewer samples than parameters
    with pytest.raises(NotImplementedError):
        L.monoexp_fit_host(np.arange(40.0), np.ones((40, 10), dtype=np.float32))  # > QMRI_MAX_ECHOES of THIS entry (the API routes such fits to qmri_lmfit_*: tests/test_lmfit_gpu.py)
    with pytest.raises(ValueError):
        L.monoexp_fit_host(x, y.astype(np.int32))  # dtype the kernel does not read
    out = L.monoexp_fit_host(x, np.zeros((4, 0), dtype=np.float32))  # empty input
    assert out["popt"].shape == (0, 2)


# ------------------------------------------------------------------------------- full size
def test_full_size_properties():
    """BASELINE config 2 size (512x512x160 x 8 echoes) through size-independent properties:
    noise-free data -> exact recovery; permutation invariance (a voxel's result does not depend on its
    position / neighbours / tile); scale equivariance a -> s*a; background -> skip rule everywhere."""
    torch = pytest.importorskip("torch")
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

    def run(yd, init, p0):
        popt = torch.empty((yd.shape[1], 2), dtype=torch.float64, device="cuda")
        r2 = torch.empty(yd.shape[1], dtype=torch.float64, device="cuda")
        a = L.default_args()
        a.y, a.y_dtype, a.E, a.N, a.ld = yd.data_ptr(), L.QMRI_F32, E, yd.shape[1], yd.shape[1]
        a.x = x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        a.init = init
        a.a0, a.b0 = p0
        a.popt, a.r2, a.out_dtype = popt.data_ptr(), r2.data_ptr(), L.QMRI_F64
        a.stream = torch.cuda.current_stream().cuda_stream
        L.check(L.load().qmri_monoexp_fit_device(ctypes.byref(a), None))
        torch.cuda.synchronize()
        return popt, r2

    popt, r2 = run(y, L.INIT_SCALAR, P0)
    fg = ~bg
    assert torch.isnan(popt[bg]).all() and (r2[bg] == 0).all()
    tc = 1 / popt[fg, 1].abs()
    # f32-rounded noise-free data -> recovery to ~1e-6.  Like lmdif itself, a handful of voxels per
    # 1e7 can stall: from p0 = (1, -1/30) the path crosses b = 0, where the forward-difference step
    # h = sqrt(eps)*|b| underflows the residual's ulp, the Jacobian's b-column cancels to exactly 0 and
    # the solver "converges" (info 2) on the flat line.  That is the reference's behaviour (oracle
    # jac_mode 0 reproduces it on such inputs), so it is bounded here, not forbidden.
    bad = ~(((tc - t2[fg]).abs() / t2[fg]) < 1e-3)
    assert bad.double().mean().item() < 1e-6, f"{int(bad.sum())} voxels off"
    ok = ~bad
    assert ((popt[fg, 0][ok] - s0[fg][ok]).abs() / s0[fg][ok]).max().item() < 1e-3
    assert (r2[fg][ok] > 0.999999).all()
    # permutation invariance on a 4M-voxel slab: bitwise identical results per voxel
    n = 1 << 22
    perm = torch.randperm(n, device="cuda", generator=gen)
    popt_p, r2_p = run(y[:, :n][:, perm].contiguous(), L.INIT_SCALAR, P0)
    a = popt[:n][perm]
    assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(popt_p, nan=-1.0))
    assert torch.equal(r2[:n][perm], r2_p)
    # scale equivariance with the log-linear init: y -> 2y gives a -> 2a, same b (exact in binary fp)
    popt1, _ = run(y[:, :n].contiguous(), L.INIT_LOGLIN, P0)
    popt2, _ = run((y[:, :n] * 2).contiguous(), L.INIT_LOGLIN, P0)
    f = fg[:n]
    assert ((popt2[f, 0] - 2 * popt1[f, 0]).abs() / popt1[f, 0].abs()).max().item() < 1e-6
    assert ((popt2[f, 1] - popt1[f, 1]).abs() / popt1[f, 1].abs()).max().item() < 1e-6


@pytest.mark.gpu
def test_kernel_exp_and_log_selftest():
    """The two elementary functions the fit kernels carry themselves (csrc/fp64_fast.h), through qmri_selftest_fp64:
    exp_sk is the device library's exp bit for bit (it is the same algorithm with its constants as scalar operands), and
    log_sk -- the logarithm of the log-linear starting point (fitting.py:701-718) -- is within 2 ulp of numpy's log over the
    whole positive range, denormals included, with log()'s values at 0, inf, NaN and negative arguments."""
    from dosma_amd import _lib as L

    rng = np.random.default_rng(5)
    x = np.concatenate([
        rng.uniform(-745.0, 710.0, 200000), rng.uniform(-4.0, 4.0, 200000), rng.standard_normal(100000) * 1e-6,
        np.array([0.0, -0.0, 709.78, 709.79, 1024.0, 1025.0, -745.2, -1075.0, -1076.0, np.inf, -np.inf, np.nan, 1e-320, 5e-324]),
        np.exp(rng.uniform(-740.0, 709.0, 300000)), rng.uniform(0.5, 2.0, 200000), rng.uniform(0.0, 4096.0, 200000),
        np.array([1.0, 2.0, 0.5, np.nextafter(1.0, 0), np.nextafter(1.0, 2), 1e-10, 2.2250738585072014e-308, 1e-310, 4.9e-324,
                  1.7976931348623157e308])])
    o = L.selftest_fp64(x)
    # exp: bit-identical to the device library, NaNs in the same places
    a, b = o["exp_sk"].view(np.uint64), o["exp_lib"].view(np.uint64)
    nan = np.isnan(o["exp_lib"])
    assert np.array_equal(np.isnan(o["exp_sk"]), nan)
    assert np.array_equal(a[~nan], b[~nan])
    # log: specials as log(), everything else within 2 ulp of numpy's (correctly rounded to < 1 ulp) value
    with np.errstate(all="ignore"):
        ref = np.log(x)
    got = o["log_sk"]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    inf = np.isinf(ref)
    assert np.array_equal(got[inf], ref[inf])
    fin = np.isfinite(ref)
    ulp = np.abs(got[fin] - ref[fin]) / np.spacing(np.abs(ref[fin]) + 5e-324)
    assert ulp.max() <= 2.0, (ulp.max(), x[fin][ulp.argmax()])
    # and the device library's own log sits in the same band (the oracle's numpy log is what both are measured against)
    ulp_lib = np.abs(o["log_lib"][fin] - ref[fin]) / np.spacing(np.abs(ref[fin]) + 5e-324)
    assert ulp_lib.max() <= 2.0


_ORDER_WORKER = r'''
import hashlib, sys
sys.path.insert(0, %(root)r)
import numpy as np
from dosma_amd import _lib as L
rng = np.random.default_rng(11)
n = 300000
x = np.arange(1, 9) * 10.0
t2 = rng.uniform(15, 80, n); s0 = rng.uniform(300, 1500, n)
y = (s0 * np.exp(-x[:, None] / t2) + rng.standard_normal((8, n)) * 15).astype(np.float32)
y[:, ::7] = 0                       # skipped voxels in between
y[:, 5::11] = rng.standard_normal((8, y[:, 5::11].shape[1])).astype(np.float32) * 40   # noise: long, failing trajectories
h = hashlib.sha1()
for p0 in ((1.0, -1 / 30.0), None):
    o = L.monoexp_fit_host(x, y, p0=p0, want_info=True) if p0 else L.monoexp_fit_host(x, y, init=L.INIT_LOGLIN, want_info=True)
    for k in ("popt", "r2", "info", "nfev"):
        h.update(np.ascontiguousarray(o[k]).tobytes())
print("SHA", h.hexdigest())
'''


@pytest.mark.gpu
def test_results_do_not_depend_on_the_order_voxels_are_pulled():
    """A voxel's result is a function of its samples alone: the lane-pull queue, the refill threshold, the result ring and
    the tile a voxel sits in only change WHEN and next to WHOM it is solved.  The same 300 000 voxels (tissue, skipped,
    pure noise) with the refill threshold at 2, 8 and 24 idle lanes -- three different interleavings -- give byte-identical
    raw outputs (a, b, r2, stop code, evaluation count), for the fixed and the log-linear start."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shas = []
    for idle in ("2", "8", "24"):
        env = dict(os.environ, QMRI_REFILL_IDLE=idle)
        p = subprocess.run([sys.executable, "-c", _ORDER_WORKER % {"root": root}], env=env, cwd=root, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        shas.append([l for l in p.stdout.splitlines() if l.startswith("SHA")][0])
    assert shas[0] == shas[1] == shas[2], shas


def test_fit_repeats_bit_for_bit():
    """The fit kernels hand out their work dynamically (waves pull tiles and refill idle lanes through atomic counters): which wave fits which
    voxel changes from launch to launch, the per-voxel results must not.  40 launches of the masked, log-linear-initialised fit on 1 Mi voxels
    (30 % background), every output array equal to the first launch's bit for bit -- the fit-side twin of
    test_unet_fullsize_gpu.py::test_512_forward_repeats_bit_for_bit (round 6: scripts/fit_repeat_check.py ran 300 launches of the bench volume
    and 50 calls of every other entry: one distinct result each, profiles/r06_repeat_sweep.txt)."""
    rng = np.random.default_rng(21)
    E, N = 8, 1 << 20
    x = np.arange(1, E + 1) * 10.0
    y = (rng.uniform(300, 1500, N) * np.exp(-x[:, None] / rng.uniform(15, 80, N)) + 18 * rng.standard_normal((E, N))).astype(np.float32)
    y[:, rng.random(N) < 0.3] = 0
    mask = rng.random(N) < 0.6
    first = None
    for rep in range(40):
        o = L.monoexp_fit_host(x, y, mask=mask, init=L.INIT_LOGLIN, want_info=True)
        got = [np.ascontiguousarray(o[k]).view(np.uint8) for k in ("popt", "r2", "info", "nfev")]
        if first is None:
            first = [g.copy() for g in got]
        else:
            assert all(np.array_equal(a, b) for a, b in zip(got, first)), f"launch {rep} differs from the first"
