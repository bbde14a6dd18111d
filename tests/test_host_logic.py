"""Host-side logic that needs no GPU: MedicalVolume / orientation shims, p0 formatting, model
recognition, argument errors -- and, when the reference is importable (build container), the shims
are compared with the reference's own classes."""
import numpy as np
import pytest

from dosma_amd import CurveFitter, MedicalVolume, MonoExponentialFit, monoexponential
from dosma_amd import fitting as F
from dosma_amd import orientation as stdo
from oracle import ref_harness

ORIENTS = [("LR", "PA", "IS"), ("SI", "AP", "LR"), ("AP", "LR", "SI"), ("RL", "IS", "PA"),
           ("IS", "RL", "AP"), ("PA", "SI", "RL")]


def test_orientation_roundtrip():
    rng = np.random.default_rng(0)
    vol = rng.random((3, 4, 5, 2))
    aff = np.diag([0.5, 0.7, 2.0, 1.0])
    aff[:3, 3] = [10.0, -20.0, 30.0]
    mv = MedicalVolume(vol, aff)
    assert mv.orientation == ("LR", "PA", "IS")
    assert np.allclose(mv.pixel_spacing, (0.5, 0.7, 2.0))
    for o in ORIENTS:
        r = mv.reformat(o)
        assert r.orientation == o
        assert np.allclose(sorted(r.pixel_spacing), [0.5, 0.7, 2.0])
        back = r.reformat(mv.orientation)
        assert np.array_equal(back.volume, vol) and np.allclose(back.affine, aff)
        # a voxel keeps its world position: index (1,2,3) of mv
        idx = np.array([1, 2, 3, 1.0])
        world = aff @ idx
        perm = stdo.get_transpose_inds(mv.orientation, o)
        j = [idx[p] for p in perm]
        for ax in range(3):
            if o[ax] != tuple(mv.orientation[p] for p in perm)[ax]:
                j[ax] = r.shape[ax] - 1 - j[ax]
        assert np.allclose(r.affine @ np.array(j + [1.0]), world)
        assert r.volume[int(j[0]), int(j[1]), int(j[2]), 0] == vol[1, 2, 3, 0]
    with pytest.raises(ValueError):
        mv.reformat(("LR", "RL", "IS"))
    assert stdo.orientation_from_affine(stdo.to_affine(("SI", "AP", "LR"), (1, 2, 3))) == ("SI", "AP", "LR")


def test_medical_volume_numpy_protocol_and_slicing():
    mv = MedicalVolume(np.arange(24.0).reshape(2, 3, 4), np.eye(4))
    assert isinstance(mv + 1, MedicalVolume) and np.array_equal((mv + 1).volume, mv.volume + 1)
    assert (mv > 3).dtype == bool and np.exp(mv).shape == mv.shape
    assert np.around(mv / 7, 2).volume[1, 1, 1] == np.around(mv.volume / 7, 2)[1, 1, 1]
    s = mv[:, 1:3, ::2]
    assert s.shape == (2, 2, 2) and np.allclose(s.affine[:3, 3], [0, 1, 0]) and s.affine[2, 2] == 2
    with pytest.raises(IndexError):
        mv[0]
    with pytest.raises(ValueError):
        mv + MedicalVolume(np.zeros((2, 3, 4)), np.diag([2.0, 1, 1, 1]))
    p = MedicalVolume(np.zeros((2, 3, 4, 2)), np.eye(4), headers=np.empty((1, 1, 4, 1), dtype=object))
    assert p[..., 1].shape == (2, 3, 4) and p[..., 1].headers().shape == (1, 1, 4)
    st = np.stack([mv, mv * 2], axis=-1)
    assert st.shape == (2, 3, 4, 2)
    assert mv.is_identical(mv.clone()) and mv.is_same_dimensions(mv[...])
    with pytest.raises(ValueError):
        MedicalVolume(np.zeros((2, 2, 2)), np.eye(4), headers=np.empty((3, 1, 1), dtype=object))


def test_format_p0_semantics():
    """reference _format_p0 (fitting.py:1106-1161)."""
    names = ["a", "b"]
    assert F._format_p0(None, names, 5) == [1.0, 1.0]
    assert F._format_p0(2.0, names, 5) == [2.0, 2.0]
    assert F._format_p0((None, 3), names, 5) == [1.0, 3.0]
    assert F._format_p0({"b": 50.0}, names, 5) == [1.0, 50.0]
    out = F._format_p0([np.ones(5), 50], names, 5)
    assert isinstance(out[0], np.ndarray) and out[1] == 50.0
    out = F._format_p0(np.ones((5, 2)), names, 5)
    assert all(isinstance(v, np.ndarray) and v.shape == (5,) for v in out)
    with pytest.raises(ValueError):
        F._format_p0((1, 2, 3), names, 5)
    with pytest.raises(ValueError):
        F._format_p0({"c": 1}, names, 5)
    with pytest.raises(ValueError):
        F._format_p0([np.ones(4), 1], names, 5)


def test_model_recognition():
    assert F._model_of(monoexponential) == "monoexponential"
    assert F._model_of(lambda x, a, b: a * np.exp(b * x)) == "monoexponential"
    assert F._model_of(lambda t, s0, r: s0 * np.exp(t * r)) == "monoexponential"
    assert F._model_of(F.biexponential) == "biexponential"
    for bad in (lambda x, a, b: a * np.exp(-b * x), lambda x, a: a * x,
                lambda x, a, b, c, d: a * np.exp(b * x) + c * np.exp(d * x),  # only the reference's own function
                lambda x, a, b: a + b * x):
        with pytest.raises(NotImplementedError):
            F._model_of(bad)


def test_constructor_validation_matches_reference():
    with pytest.raises(ValueError):
        MonoExponentialFit(tc0="a value")
    with pytest.raises(ValueError):
        MonoExponentialFit(bounds=(0, 1, 2))
    with pytest.raises(ValueError):
        CurveFitter(monoexponential, out_bounds=[(0, 0.5, 1.0)])
    with pytest.raises(ValueError):
        CurveFitter(monoexponential, out_bounds=[(1.2, 0)])
    with pytest.raises(TypeError):
        CurveFitter(monoexponential, out_ufuncs=[None, 5])
    with pytest.raises(ValueError):
        CurveFitter(monoexponential, r2_threshold="nope")
    assert CurveFitter(monoexponential).r2_threshold == 0.9  # "preferences" -> fitting/r2.threshold
    f = CurveFitter(monoexponential, out_ufuncs=(None, F._inv_abs), out_bounds=((-np.inf, np.inf), (0, 100)),
                    nan_to_num=0.0)
    post = f._fusable_post()
    assert post["inv_abs_b"] and post["bounds"] == ((-np.inf, np.inf), (0.0, 100.0))
    assert CurveFitter(monoexponential, out_ufuncs=lambda v: v)._fusable_post() is None
    y = [MedicalVolume(np.ones((2, 2, 2)), np.eye(4)) for _ in range(4)]
    with pytest.raises(TypeError):
        MonoExponentialFit().fit([1, 2, 3, 4], [v.A for v in y])
    with pytest.raises(ValueError):
        MonoExponentialFit().fit([1, 2, 3], y)


def _gauss(t, a, mu, sig):
    return a * np.exp(-0.5 * ((t - mu) / sig) ** 2)


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference absent (GPU box)")
def test_requests_the_kernels_do_not_implement_follow_the_reference():
    """SURVEY 8(b)'s dispatch rule, second half: a generic `func`, or scipy kwargs that select another solver, run the
    reference's behaviour -- one scipy.optimize.curve_fit per voxel under its skip / failure / r2 rules
    (fitting.py:318-321, 827-870, 1026-1073).  Compared with the live reference on the same inputs: equal, not close
    (it is the same scipy call on the same numbers).  No GPU involved."""
    import warnings

    import dosma_amd as dm

    dosma = ref_harness.load_reference()
    rng = np.random.default_rng(11)
    x = np.linspace(-2.0, 3.0, 12)
    N = 40
    truth = np.stack([rng.uniform(1, 5, N), rng.uniform(-0.5, 1.0, N), rng.uniform(0.6, 1.5, N)], axis=1)
    y = np.stack([_gauss(x, *p) for p in truth], axis=1) + 0.02 * rng.standard_normal((12, N))
    y[:, 3] = 0                       # skip rule
    y[:, 5] = 1e3                     # flat far-away column: the solver hits maxfev or converges badly -- whatever scipy does
    p0 = (2.0, 0.0, 1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # (a) a 3-parameter user model, scalar p0, y_bounds
        ref_p, ref_r = dosma.curve_fit(_gauss, x, y, p0=p0, y_bounds=(-10, 100))
        got_p, got_r = dm.curve_fit(_gauss, x, y, p0=p0, y_bounds=(-10, 100))
        assert np.array_equal(got_p, ref_p, equal_nan=True) and np.array_equal(got_r, ref_r)
        assert np.isnan(got_p[3]).all() and got_r[3] == 0 and np.isnan(got_p[5]).all()  # all-zero / out of y_bounds
        assert np.abs(got_p[6:] - truth[6:]).max() < 0.2
        # (b) per-voxel p0 for one parameter, maxfev small enough that some voxels fail -> (nan, ...), 0
        p0v = {"a": truth[:, 0] * 1.3, "mu": 0.0, "sig": 1.0}
        ref_p, ref_r = dosma.curve_fit(_gauss, x, y, p0=p0v, maxfev=25)
        got_p, got_r = dm.curve_fit(_gauss, x, y, p0=p0v, maxfev=25)
        assert np.array_equal(got_p, ref_p, equal_nan=True) and np.array_equal(got_r, ref_r)
        assert np.isnan(got_p[:, 0]).sum() > 1 and np.isfinite(got_p[:, 0]).sum() > 10
        # (c) the kernels' own model with bounds= : scipy switches to trf and counts max_nfev (fitting.py:827-830)
        te = np.arange(1, 9) * 10.0
        ym = np.stack([s0 * np.exp(-te / t2) for s0, t2 in zip(rng.uniform(300, 1500, N), rng.uniform(15, 80, N))], axis=1)
        ym += 10 * rng.standard_normal(ym.shape)
        kw = dict(p0=(1000.0, -0.03), bounds=([0.0, -1.0], [5000.0, 0.0]))
        ref_p, ref_r = dosma.curve_fit(dosma.monoexponential, te, ym, **kw)
        got_p, got_r = dm.curve_fit(dm.monoexponential, te, ym, **kw)
        assert np.array_equal(got_p, ref_p, equal_nan=True) and np.array_equal(got_r, ref_r)
        # (d) sigma= through CurveFitter on MedicalVolumes with a mask, out_bounds, r2 threshold, nan_to_num
        shape = (5, 4, 2)
        vols = [dm.MedicalVolume(ym[e].reshape(shape), np.eye(4)) for e in range(8)]
        rvols = [dosma.MedicalVolume(ym[e].reshape(shape), np.eye(4)) for e in range(8)]
        mask = (np.arange(N).reshape(shape) % 3 != 0)
        opts = dict(p0=(1000.0, -0.03), out_bounds=((0, 1400), (-1, 0)), r2_threshold=0.5, nan_to_num=-1.0,
                    sigma=np.linspace(1.0, 2.0, 8))
        ref_pv, ref_rv = dosma.CurveFitter(dosma.monoexponential, **opts).fit(te, rvols, mask=dosma.MedicalVolume(mask, np.eye(4)))
        got_pv, got_rv = dm.CurveFitter(dm.monoexponential, **opts).fit(te, vols, mask=dm.MedicalVolume(mask, np.eye(4)))
        assert np.array_equal(got_pv.volume, ref_pv.volume) and np.array_equal(got_rv.volume, ref_rv.volume)
        assert (got_pv.volume[~mask] == -1.0).all() and (got_pv.volume == -1.0).sum() > (~mask).sum() * 2  # some out of bounds
        # (e) a generic 1-parameter func through CurveFitter, default p0 (scipy: ones), process pool like the reference's
        lin = lambda t, a: a * t  # noqa: E731
        ref_pv, ref_rv = dosma.CurveFitter(lin, r2_threshold=None).fit(te, rvols)
        got_pv, got_rv = dm.CurveFitter(lin, r2_threshold=None).fit(te, vols)
        assert np.array_equal(got_pv.volume, ref_pv.volume) and np.array_equal(got_rv.volume, ref_rv.volume)
        got_p, got_r = dm.curve_fit(_gauss, x, y, p0=p0, num_workers=2, chunksize=7)
        ref_p, ref_r = dosma.curve_fit(_gauss, x, y, p0=p0)
        assert np.array_equal(got_p, ref_p, equal_nan=True) and np.array_equal(got_r, ref_r)
    # the route announces itself (a per-voxel CPU loop is not what a caller of this package expects silently)
    with pytest.warns(RuntimeWarning, match="per-voxel scipy"):
        dm.curve_fit(lin, te, ym[:, :2])


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference absent (GPU box)")
def test_medical_volume_shim_vs_reference():
    dosma = ref_harness.load_reference()
    rng = np.random.default_rng(3)
    vol = rng.random((3, 4, 5))
    aff = np.array([[0, 0, 0.8, 5.0], [-0.6, 0, 0, 7.0], [0, -1.5, 0, -3.0], [0, 0, 0, 1.0]])
    ours, ref = MedicalVolume(vol, aff), dosma.MedicalVolume(vol, aff)
    assert ours.orientation == ref.orientation
    for o in ORIENTS:
        a, b = ours.reformat(o), ref.reformat(o)
        assert a.orientation == b.orientation == o
        assert np.array_equal(a.volume, b.volume) and np.allclose(a.affine, b.affine)
    a, b = ours[1:, ::2, 1:4], ref[1:, ::2, 1:4]
    assert np.array_equal(a.volume, b.volume) and np.allclose(a.affine, b.affine)


def test_quantitative_values_to_metrics():
    """reference tests/core/test_quant_vals.py:52-174: exact counts / means per label, bounds `closed=`."""
    from dosma_amd.quant_vals import QuantitativeValue, QuantitativeValueType, T1Rho, T2, T2Star

    vol = np.zeros((4, 4, 2))
    vol[:2] = 10.0
    vol[2:] = 30.0
    vol[0, 0, 0] = np.nan
    vol[3, 3, 1] = 100.0
    qv = T2(MedicalVolume(vol, np.eye(4)))
    assert qv.qv_type == QuantitativeValueType.T2 and QuantitativeValue.get_qv("t2").NAME == "t2"
    assert isinstance(QuantitativeValue.get_qv(1), T1Rho) and isinstance(QuantitativeValue.get_qv("t2_star"), T2Star)
    with pytest.raises(ValueError):
        QuantitativeValue.get_qv("nope")
    with pytest.raises(TypeError):
        T2(vol)
    df = qv.to_metrics()
    assert list(df["Category"]) == ["total"] and df["# Voxels"][0] == 31
    labels = np.zeros((4, 4, 2), dtype=np.uint8)
    labels[:2] = 1
    labels[2:] = 2
    df = qv.to_metrics(MedicalVolume(labels, np.eye(4)), labels={1: "a", 2: "b"})
    assert list(df["Category"]) == ["a", "b", "total"]
    assert list(df["# Voxels"]) == [15, 16, 31] and df["Mean"][0] == 10.0
    df = qv.to_metrics(MedicalVolume(labels, np.eye(4)), bounds=(10, 30), closed="right")
    assert list(df["# Voxels"]) == [0, 15, 15]  # 10 excluded (open on the left), 100 excluded
    df = qv.to_metrics(MedicalVolume(labels, np.eye(4)), bounds=(10, 30), closed="both",
                       fns={"max": lambda v: np.max(v) if v.size else np.nan})
    assert list(df["# Voxels"]) == [15, 15, 30] and df["max"][1] == 30.0
    qv.add_additional_volume("r2", MedicalVolume(np.ones((4, 4, 2)), np.eye(4)))
    assert "r2" in qv.additional_volumes
    with pytest.raises(TypeError):
        qv.add_additional_volume("r2", np.ones(3))


G8_CALLS = {
    "nomask": lambda qv, m: qv.to_metrics(),
    "auto": lambda qv, m: qv.to_metrics(m),
    "subset": lambda qv, m: qv.to_metrics(m, labels={3: "tc", 1: "fc"}),
    "b_right": lambda qv, m: qv.to_metrics(m, bounds=(20.0, 60.0), closed="right"),
    "b_left": lambda qv, m: qv.to_metrics(m, bounds=(20.0, 60.0), closed="left"),
    "b_both": lambda qv, m: qv.to_metrics(m, bounds=(20.0, 60.0), closed="both"),
    "b_neither": lambda qv, m: qv.to_metrics(m, bounds=(20.0, 60.0), closed="neither"),
    "b_nomask": lambda qv, m: qv.to_metrics(bounds=(20.0, 60.0)),
    "empty": lambda qv, m: qv.to_metrics(m, labels={4: "men"}, bounds=(1000.0, 2000.0)),
}


def check_against_g8(g, tag, case, df, exact_moments):
    """One DataFrame against the real reference's (tests/golden/g8_to_metrics.npz, made by oracle/make_golden.py g8
    from /root/reference/dosma/core/quant_vals.py:145-229).  Counts and medians are exact; means / standard deviations
    are exact when the evaluation is numpy's own (host route), else fp64 sums in another order."""
    key = f"{tag}_{case}"
    assert list(df["Category"]) == list(g[f"{key}_category"]), key
    assert np.array_equal(df["# Voxels"].to_numpy(np.float64), g[f"{key}_count"]), key
    f32 = tag == "float32"
    for col, name in (("Median", "median"), ("Mean", "mean"), ("Std", "std")):
        ours, ref = df[col].to_numpy(np.float64), g[f"{key}_{name}"]
        assert np.array_equal(np.isnan(ours), np.isnan(ref)), (key, col)
        ok = ~np.isnan(ref)
        if name == "median" or exact_moments:
            assert np.array_equal(ours[ok], ref[ok]), (key, col, ours, ref)
        else:  # float32 maps: numpy (the reference) accumulates in float32, the kernel in float64
            tol = 2e-6 if f32 else 1e-12
            assert np.all(np.abs(ours[ok] - ref[ok]) <= tol * np.maximum(1.0, np.abs(ref[ok]))), (key, col, ours, ref)


@pytest.mark.parametrize("tag", ["float64", "float32"])
def test_to_metrics_host_route_vs_reference_golden(golden, tag):
    """The host (numpy) route of to_metrics -- taken whenever `fns` holds user callables -- against the reference's own
    output on the same seeded map: every column bit-equal."""
    from dosma_amd.quant_vals import T2

    g = golden("g8_to_metrics.npz")
    qv = T2(MedicalVolume(g["vol"].astype(tag), np.eye(4)))
    mask = MedicalVolume(g["labels"], np.eye(4))
    for case in G8_CALLS:
        kw_fns = {"n": lambda v: v.size}
        calls = {
            "nomask": lambda: qv.to_metrics(fns=kw_fns),
            "auto": lambda: qv.to_metrics(mask, fns=kw_fns),
            "subset": lambda: qv.to_metrics(mask, labels={3: "tc", 1: "fc"}, fns=kw_fns),
            "b_nomask": lambda: qv.to_metrics(bounds=(20.0, 60.0), fns=kw_fns),
            "empty": lambda: qv.to_metrics(mask, labels={4: "men"}, bounds=(1000.0, 2000.0), fns=kw_fns),
        }
        if case.startswith("b_") and case != "b_nomask":
            df = qv.to_metrics(mask, bounds=(20.0, 60.0), closed=case[2:], fns=kw_fns)
        else:
            df = calls[case]()
        check_against_g8(g, tag, case, df.drop(columns=["n"]), exact_moments=True)


def test_bench_bookkeeping_helpers():
    """bench.py's constants that the judge recomputes: the layer-by-layer minimum HBM traffic of a 160-slice forward as built
    (25.4 GB written + 25.5 GB read), the source hashes the profile constants are guarded with, the cgroup quota."""
    import bench

    total, wr, rd = bench.unet_algorithmic_bytes()
    assert abs(wr / 1e9 - 25.34) < 0.05 and abs(rd / 1e9 - 25.53) < 0.05 and total == wr + rd
    t512, _, _ = bench.unet_algorithmic_bytes(hw=512)
    assert 1.7 < t512 / total < 1.8                        # ~(512 / 384)^2; which levels pool in a kernel of their own differs
    assert len(bench._kernel_source_sha1()) == 40 and bench._kernel_source_sha1() != bench._unet_source_sha1()
    q = bench.effective_cores()
    assert q is None or q > 0
    tr = bench._unet_traffic()
    assert tr["algorithmic_bytes"] == total and ("traffic_source" not in tr or "stale" in tr["traffic_source"])


def test_monoexponential_fit_with_more_samples_than_the_kernels_keep(golden):
    """with 65 samples per voxel (> 64, the general lmdif kernel's limit) MonoExponentialFit.fit goes through the
    per-voxel scipy loop and must still return what the reference returns (fitting.py:720-744): the ROUNDED tc map (not the
    unrounded (a, tc) pairs) and r2.  Golden g0 = the real reference on the same inputs; same scipy call on the same numbers,
    so equal.  The plain CurveFitter with a mask through the same route as well.  No GPU involved."""
    import warnings

    g = golden("g0_many_samples.npz")
    x, y, mask = g["x"], g["y"], g["mask"]
    vols = [MedicalVolume(v, np.eye(4)) for v in y]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tc, r2 = MonoExponentialFit(decimal_precision=6).fit(x, vols)
        popt, r2m = CurveFitter(monoexponential, p0=(1.0, -1 / 30.0)).fit(x, vols, mask=mask)
    assert tc.shape == y.shape[1:] and r2.shape == y.shape[1:]
    np.testing.assert_allclose(tc.volume, g["tc_default"], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(r2.volume, g["r2_default"], rtol=1e-9, atol=1e-12)
    assert tc.volume[0, 0, 0] == 0 and (tc.volume > 0).sum() > 100
    np.testing.assert_allclose(popt.volume, g["popt_masked"], rtol=1e-9, equal_nan=True)
    np.testing.assert_allclose(r2m.volume, g["r2_masked"], rtol=1e-9, atol=1e-12, equal_nan=True)


def test_bench_gpu_sampler_never_raises_without_a_gpu():
    """bench.py's clock / power sampler (hwmon files of the device's PCI function, else rocm-smi) on a box with neither: it starts, stops
    and summarises to "no samples" -- the bench line then carries nulls, it does not fail."""
    import time

    import torch

    import bench

    sp = bench.GpuSampler(torch, 0)
    with sp:
        time.sleep(0.02)
    out = sp.summary()
    assert set(out) >= {"samples", "how", "sclk_ghz_mean", "power_w_mean"}
    assert out["sclk_ghz_mean"] is None or out["sclk_ghz_mean"] > 0
    assert bench.effective_cores() is None or bench.effective_cores() > 0


def test_bench_cpu_baseline_reports_the_fastest_rung():
    """bench.py's cpu_baseline on a small CPU tensor (no GPU): the reference's call pattern through a worker ladder, `value` = the faster of the
    ladder's two best rungs timed on the whole sample, the worker count stated (Pool(256) on a 16-core quota cost the
    baseline 25 %)."""
    import torch

    import bench

    rng = np.random.default_rng(2)
    n = 3000
    y = (rng.uniform(300, 1500, n) * np.exp(-bench.TE[:, None] / rng.uniform(15, 80, n)) + 18 * rng.standard_normal((bench.E, n))).astype(np.float32)
    y[:, ::4] = 0
    out = bench.cpu_baseline(torch.from_numpy(y), cores_cap=2)
    assert out["kind"] == "port" and out["unit"] == "voxel-fits/s" and out["value"] > 100
    assert out["workers"] == out["cores"] and out["workers"] in (1, 2) and out["visible_cores"] == 2
    assert set(out["worker_ladder_voxel_fits_per_s"]) == {"1", "2"}
    assert str(out["workers"]) in out["full_sample_voxel_fits_per_s_by_workers"]
    assert out["value"] == max(out["full_sample_voxel_fits_per_s_by_workers"].values())
    assert f"Pool({out['workers']})" in out["sample"]
