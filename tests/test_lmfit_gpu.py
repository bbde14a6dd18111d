"""GPU parity of the general lmdif kernel (lm_generic.hip) through the C ABI:
bi-exponential fits vs golden vectors made by the real reference (g7), and the mono-exponential model
with true forward differences vs the reference golden (g2) and vs the fast kernel's emulated differences.
Tolerance: 1e-4 relative (north_star)."""
import numpy as np
import pytest

import dosma_amd as dm
from dosma_amd import _lib as L

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def test_biexponential_vs_reference_golden(golden, relerr):
    g = golden("g7_biexp.npz")
    x, y, p0 = g["x"], g["y"], tuple(g["p0"])
    o = L.lmfit_host("biexponential", x, y, p0, want_info=True)
    ok_ref = ~np.isnan(g["popt"][:, 0])
    ok = (o["info"] >= 1) & (o["info"] <= 4)
    assert (ok == ok_ref).mean() > 0.995          # noise voxels may flip class; tissue must not
    tissue = np.arange(y.shape[1]) >= 60
    assert (ok == ok_ref)[tissue].all()
    assert (o["info"][:20] == 0).all() and np.isnan(o["popt"][:20]).all() and (o["r2"][:20] == 0).all()
    both = ok & ok_ref & tissue
    d = relerr(o["popt"][both], g["popt"][both]).max(axis=1)
    assert d.max() < RTOL, f"max rel {d.max()}"
    assert np.abs(o["r2"][both] - g["r2"][both]).max() < 1e-6
    assert (o["nfev"][both] == g["nfev"][both]).mean() > 0.95
    assert np.isnan(o["popt"][~ok]).all() and (o["r2"][~ok] == 0).all()


def test_biexponential_curve_fit_api(golden, relerr):
    """dosma.curve_fit(biexponential, x, y32, p0=dict with per-voxel arrays, y_bounds=...)."""
    g = golden("g7_biexp.npz")
    x, y32 = g["x"], g["y32"]
    p0 = {"a1": g["p0v_a1"], "b1": -0.1, "a2": 400.0, "b2": g["p0v_b2"]}
    popt, r2 = dm.curve_fit(dm.biexponential, x, y32, p0=p0, y_bounds=(-50, 1500))
    ref = g["popt_f32"]
    assert popt.shape == ref.shape and popt.dtype == np.float64
    tissue = np.arange(y32.shape[1]) >= 60
    cls = np.isnan(popt[:, 0]) == np.isnan(ref[:, 0])
    assert cls[tissue].mean() > 0.999 and cls.mean() > 0.99
    both = tissue & ~np.isnan(popt[:, 0]) & ~np.isnan(ref[:, 0])
    d = relerr(popt[both], ref[both]).max(axis=1)
    assert (d < RTOL).mean() > 0.998, f"{(d >= RTOL).sum()} of {both.sum()} beyond 1e-4"
    # default p0 = ones: mostly garbage/failures in the reference too; the failure class must agree
    popt1, _ = dm.curve_fit(dm.biexponential, x, g["y"][:, :300])
    assert (np.isnan(popt1[:, 0]) == np.isnan(g["popt_ones"][:, 0])).mean() > 0.97
    with pytest.raises(TypeError):
        dm.curve_fit(dm.biexponential, x[:3], g["y"][:3, :10])
    with pytest.raises(ValueError):
        dm.curve_fit(dm.biexponential, x, g["y"], p0=(1.0, 2.0))


def test_biexponential_curvefitter_volumes(golden, relerr):
    """CurveFitter(biexponential, out_ufuncs, out_bounds, r2_threshold, nan_to_num).fit(x, vols, mask)."""
    g = golden("g7_biexp.npz")
    x, y, mask = g["x"], g["y"], g["mask"]
    shape = mask.shape
    vols = [dm.MedicalVolume(np.ascontiguousarray(v.reshape(shape)), np.eye(4)) for v in y]

    def inv(v):
        return 1 / np.abs(v)

    cf = dm.CurveFitter(dm.biexponential, p0=tuple(g["p0"]), out_ufuncs=[None, inv, None, inv],
                        out_bounds=(0, 2000), r2_threshold=0.9, nan_to_num=0.0)
    pm, rm = cf.fit(x, vols, mask=dm.MedicalVolume(mask.astype(np.uint8), np.eye(4)))
    assert pm.shape == shape + (4,) and rm.shape == shape
    ref, rref = g["popt_cf"], g["r2_cf"]
    assert (pm.A[~mask] == 0).all() and (rm.A[~mask] == 0).all()
    flat = np.arange(mask.size).reshape(shape)
    tissue = mask & (flat >= 60)
    zero_cls = (pm.A[tissue][:, 0] == 0) == (ref[tissue][:, 0] == 0)
    assert zero_cls.mean() > 0.995
    d = relerr(pm.A[tissue], ref[tissue]).max(axis=1)
    assert (d < RTOL).mean() > 0.995
    assert np.abs(rm.A[tissue] - rref[tissue]).max() < 1e-6


@pytest.mark.parametrize("snr", [100, 20])
def test_true_forward_differences_monoexp_vs_golden_and_fast_kernel(golden, relerr, snr):
    """The same lmdif with n = 2: (1) vs the reference golden g2, (2) vs the fast kernel, whose forward
    differences are emulated without extra exponentials (monoexp_lm.hip) -- same decisions, same answers."""
    g = golden("g2_cfg2_8echo.npz")
    x, y = g["x"], g[f"y_snr{snr}"]
    p0 = (1.0, -1 / 30.0)
    o = L.lmfit_host("monoexponential", x, y, p0, want_info=True)
    f = L.monoexp_fit_host(x, y, p0=p0, want_info=True)
    assert relerr(o["popt"], g[f"popt_snr{snr}"]).max() < RTOL
    assert (o["nfev"] == g[f"nfev_snr{snr}"]).mean() > 0.999
    assert (o["info"] == f["info"]).mean() > 0.999 and (o["nfev"] == f["nfev"]).mean() > 0.999
    assert relerr(o["popt"], f["popt"]).max() < RTOL
    assert np.abs(o["r2"] - f["r2"]).max() < 1e-6


def test_lmfit_edges():
    x = np.linspace(4.0, 92.0, 12)
    y = np.zeros((12, 130), np.int16)
    y[:, 1] = (900 * np.exp(-x / 10) + 500 * np.exp(-x / 70)).astype(np.int16)
    o = L.lmfit_host("biexponential", x, y, (500.0, -0.1, 500.0, -0.02), want_info=True)
    assert (o["info"][[0, 2, 129]] == 0).all() and 1 <= o["info"][1] <= 4
    yb = y.astype(np.float64)
    yb[3, 5] = np.nan
    with pytest.raises(ValueError):
        L.lmfit_host("biexponential", x, yb, (500.0, -0.1, 500.0, -0.02))
    e = L.lmfit_host("biexponential", x, np.zeros((12, 0)), (1.0, 1.0, 1.0, 1.0))
    assert e["popt"].shape == (0, 4)


@pytest.mark.parametrize("model,E", [("biexponential", 5), ("biexponential", 8), ("biexponential", 10),
                                     ("biexponential", 12), ("biexponential", 16),
                                     ("monoexponential", 3), ("monoexponential", 8), ("monoexponential", 11),
                                     # beyond the 32 samples of the register-resident kernels: the general-E
                                     # kernel holds (n + 3) E doubles per lane in LDS -- 64 samples for 2 parameters, 45 for 4
                                     ("monoexponential", 40), ("monoexponential", 64), ("biexponential", 40)])
def test_every_echo_count_variant_vs_oracle(relerr, model, E):
    """The pulling kernel is instantiated per echo-count class (E = 8 and E = 12 unrolled exactly, E < 8 and 8 < E < 12
    with guards) and E > 12 runs the general-E kernel: each against the C restatement of lmdif (oracle/minpack_oracle.c),
    seeded two-compartment data with background (zeros) and a few voxels that exhaust maxfev."""
    import oracle.fit_oracle as fo
    rng = np.random.default_rng(100 + E)
    n = 3000
    x = np.linspace(4.0, 92.0, E)
    a1, a2 = rng.uniform(300, 900, n), rng.uniform(200, 700, n)
    ts, tl = rng.uniform(5, 15, n), rng.uniform(40, 90, n)
    y = a1 * np.exp(-x[:, None] / ts) + a2 * np.exp(-x[:, None] / tl) + 2.0 * rng.standard_normal((E, n))
    y[:, :200] = 0
    y = np.ascontiguousarray(y.astype(np.float32))
    p0 = (500.0, -0.1, 500.0, -0.02) if model == "biexponential" else (1.0, -1 / 30.0)
    o = L.lmfit_host(model, x, y, p0, want_info=True)
    popt, r2, info, nfev = fo.curve_fit_c(x, y, p0=p0, model=model, full_output=True)
    ok_ref = ~np.isnan(popt[:, 0])
    ok = (o["info"] >= 1) & (o["info"] <= 4)
    assert (o["info"][:200] == 0).all() and np.isnan(o["popt"][:200]).all()
    assert (ok == ok_ref).mean() > 0.995
    both = ok & ok_ref
    assert both.sum() > 1000
    d = relerr(o["popt"][both], popt[both]).max(axis=1)
    # the 4-parameter problem is ill-conditioned: a last-ulp difference in one division can end a fit one evaluation apart
    assert (d < RTOL).mean() > (0.99 if model == "biexponential" else 0.9999), f"{(d >= RTOL).sum()} of {both.sum()}"
    assert (o["nfev"][both] == nfev[both]).mean() > (0.95 if model == "biexponential" else 0.999)
    assert np.abs(o["r2"][both] - r2[both])[d < RTOL].max() < 1e-6


def test_more_than_32_samples_through_the_api(relerr):
    """The reference has no limit on the samples per voxel (/root/reference/dosma/core/fitting.py:755-870).  Beyond the 32 the
    mono-exponential kernel keeps in registers, curve_fit / CurveFitter / MonoExponentialFit run on the general lmdif kernel:
    against the per-voxel scipy loop of the oracle at E = 40 and 64, incl. a mask, the polyfit start and the rounded map."""
    import dosma_amd as dm
    import oracle.fit_oracle as fo
    rng = np.random.default_rng(64)
    for E in (40, 64):
        x = np.linspace(2.0, 120.0, E)
        shape = (6, 5, 4)
        n = int(np.prod(shape))
        s0, t2 = rng.uniform(400, 1500, n), rng.uniform(15, 80, n)
        y = s0 * np.exp(-x[:, None] / t2) + 8.0 * rng.standard_normal((E, n))
        y[:, :10] = 0
        y = y.astype(np.float32)
        popt, r2 = dm.curve_fit(dm.monoexponential, x, y, p0=(1.0, -1 / 30.0))
        ref_popt, ref_r2 = fo.curve_fit_scipy(x, y, p0=(1.0, -1 / 30.0))
        ok = ~np.isnan(ref_popt[:, 0])
        assert np.array_equal(np.isnan(popt[:, 0]), ~ok) and ok.sum() > 100
        assert relerr(popt[ok], ref_popt[ok]).max() < RTOL and np.abs(r2[ok] - ref_r2[ok]).max() < 1e-6
        vols = [dm.MedicalVolume(y[e].reshape(shape), np.eye(4)) for e in range(E)]
        mask = np.zeros(shape, bool)
        mask[1:5] = True
        for tc0 in (30.0, "polyfit"):
            tc, r2v = dm.MonoExponentialFit(tc0=tc0, decimal_precision=3).fit(x, vols, mask=dm.MedicalVolume(mask.astype(np.uint8), np.eye(4)))
            tc_ref, r2_ref, _ = fo.monoexp_fit_arrays(x, y, mask=mask.reshape(-1), tc0=tc0, decimal_precision=3)
            bad = np.abs(tc.volume.reshape(-1) - tc_ref) > 1e-3 + 1e-9
            assert bad.mean() < 0.01, (E, tc0, int(bad.sum()))
            assert np.abs(r2v.volume.reshape(-1) - r2_ref).max() < 1e-5
            assert (tc.volume[0] == 0).all() and (tc.volume[5] == 0).all()
    # beyond 64 samples per voxel the kernels do not go: the reference's own per-voxel scipy loop (it has no limit)
    xs = np.arange(1.0, 66.0)
    with pytest.warns(RuntimeWarning, match="per-voxel scipy"):
        pl, rl = dm.curve_fit(dm.monoexponential, xs, np.outer(np.exp(-0.05 * xs), [1.0, 3.0]).astype(np.float32), p0=(1.0, -0.1))
    assert np.allclose(pl, [[1.0, -0.05], [3.0, -0.05]], rtol=1e-4) and (rl > 0.999999).all()
