"""Generate tests/golden/*.npz by running the REAL reference (ad12/DOSMA @ /root/reference).

Build-side tool, this container only (see oracle/ref_harness.py).  The reference's own tests hold no
stored vectors for the fit path (all its data is unseeded random), so the
golden vectors are produced here from seeded inputs by calling the reference's public API:
``dosma.curve_fit``, ``dosma.CurveFitter``, ``dosma.MonoExponentialFit`` (dosma/core/fitting.py).
A fixture is data only: inputs + the reference's outputs (+ scipy's ier/nfev for the same call).

    python oracle/make_golden.py [g0 ... g9]   # writes tests/golden/g*.npz  (~30 s, 8 workers)
"""
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
# QMRI_GOLDEN_OUT: write somewhere else (tests/test_oracle.py::test_fixture_recipes_reproduce_the_committed_fixtures
# regenerates every fixture into a temp dir and compares it with the committed one)
OUT = os.environ.get("QMRI_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")

from oracle.ref_harness import load_reference  # noqa: E402

dosma = load_reference()
MV = dosma.MedicalVolume
NW = min(8, os.cpu_count() or 1)


def vols(y3d):
    """list of MedicalVolumes from (E, X, Y, Z)."""
    return [MV(np.array(v), affine=np.eye(4)) for v in y3d]


def scipy_info(x, y, p0, func=None):
    """ier / nfev of the very call the reference makes (fitting.py:1030), per voxel."""
    func = func or dosma.monoexponential
    from scipy import optimize as sop

    ier = np.zeros(y.shape[1], np.int32)
    nfev = np.zeros(y.shape[1], np.int32)
    for i in range(y.shape[1]):
        yi = y[:, i]
        if (yi == 0).all():
            continue
        p0i = tuple(float(v[i]) if isinstance(v, np.ndarray) else float(v) for v in p0)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out = sop.curve_fit(func, x, yi, p0=p0i, ftol=1e-5, maxfev=100,
                                    full_output=True)
            ier[i], nfev[i] = out[4], out[2]["nfev"]
        except RuntimeError:
            ier[i], nfev[i] = 5, -1
    return ier, nfev


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


def g1():
    """The reference tests' own generator, seeded (tests/core/test_fitting.py:18-31, 199-277, 283-291)."""
    rng = np.random.default_rng(1)
    shape = (10, 10, 20)
    x = np.asarray([0.5, 1.0, 2.0, 4.0])
    b = rng.random(shape) + 0.1
    y = np.stack([dosma.monoexponential(t, 1.0, b) for t in x])  # float64
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tc_def, r2_def = dosma.MonoExponentialFit(decimal_precision=8).fit(x, vols(y))
        tc_pf, r2_pf = dosma.MonoExponentialFit(tc0="polyfit", decimal_precision=8).fit(x, vols(y))
        popt, r2 = dosma.CurveFitter(dosma.monoexponential).fit(x, vols(y))
        mask = rng.random(shape) > 0.5
        popt_m, r2_m = dosma.CurveFitter(dosma.monoexponential).fit(x, vols(y), mask=mask)
        tc_m, r2_tm = dosma.MonoExponentialFit(decimal_precision=8).fit(x, vols(y), mask)
        # zeros in echo 0 (test_fitting.py:267-277)
        y0 = y.copy()
        y0[0, :5, :5] = 0
        tc_z, r2_z = dosma.MonoExponentialFit(tc0="polyfit", decimal_precision=8).fit(x, vols(y0))
    save("g1_tests_generator.npz", x=x, b=b, y=y, tc_default=tc_def.A, r2_default=r2_def.A,
         tc_polyfit=tc_pf.A, r2_polyfit=r2_pf.A, popt=popt.A, r2=r2.A, mask=mask,
         popt_masked=popt_m.A, r2_masked=r2_m.A, tc_masked=tc_m.A, r2_tc_masked=r2_tm.A,
         y_zero_echo0=y0, tc_zero_echo0=tc_z.A, r2_zero_echo0=r2_z.A)


def cfg2_like(rng, n_tissue, n_bg, snr, E=8):
    """SURVEY.md 8(d) cfg2 distribution: S0~U(300,1500), T2~U(15,80) ms, TE=10..80, sigma=mean(S0)/SNR."""
    x = np.arange(1, E + 1) * 10.0
    s0 = rng.uniform(300, 1500, n_tissue)
    t2 = rng.uniform(15, 80, n_tissue)
    y = s0 * np.exp(-x[:, None] / t2) + (900.0 / snr) * rng.standard_normal((E, n_tissue))
    y = np.concatenate([y, np.zeros((E, n_bg))], axis=1).astype(np.float32)
    return x, y


def g2():
    """8-echo noisy set (headline config), fixed p0 and polyfit init, three SNRs."""
    rng = np.random.default_rng(20260928)
    out = {}
    for snr in (100, 50, 20):
        x, y = cfg2_like(rng, 4000, 200, snr)
        N = y.shape[1]
        # raw curve_fit, p0 = MonoExponentialFit's default (1, -1/30)   (fitting.py:720)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            popt, r2 = dosma.curve_fit(dosma.monoexponential, x, y, p0=(1.0, -1 / 30.0),
                                       num_workers=NW, chunksize=500)
            ier, nfev = scipy_info(x, y, (1.0, -1 / 30.0))
            # full recipes on a (N,1,1) volume
            y3 = y.reshape(y.shape[0], N, 1, 1)
            tc_a, r2_a = dosma.MonoExponentialFit(num_workers=NW).fit(x, vols(y3))
            tc_b, r2_b = dosma.MonoExponentialFit(tc0="polyfit", decimal_precision=3,
                                                  num_workers=NW).fit(x, vols(y3))
        out.update({f"y_snr{snr}": y, f"popt_snr{snr}": popt, f"r2_snr{snr}": r2,
                    f"ier_snr{snr}": ier, f"nfev_snr{snr}": nfev,
                    f"tcA_snr{snr}": tc_a.A.reshape(-1), f"r2A_snr{snr}": r2_a.A.reshape(-1),
                    f"tcB_snr{snr}": tc_b.A.reshape(-1), f"r2B_snr{snr}": r2_b.A.reshape(-1)})
    save("g2_cfg2_8echo.npz", x=x, **out)


def g3():
    """Edge cases: skip rule, negatives, pure noise, flat signal, maxfev exhaustion, far p0, int input."""
    rng = np.random.default_rng(3)
    E = 8
    x = np.arange(1, E + 1) * 10.0
    cols = []
    cols.append(np.zeros(E))                                   # all zero -> skipped
    cols.append(np.r_[0.0, 500 * np.exp(-x[1:] / 40)])         # zero only in echo 0
    cols.append(-500 * np.exp(-x / 40))                        # negative signal
    cols.append(np.full(E, 700.0))                             # flat (b -> 0)
    cols.append(500 * np.exp(+x / 60))                         # growing
    cols.append(np.r_[1000.0, np.zeros(E - 1)])                # spike
    cols.append(1e-3 * np.exp(-x / 25))                        # tiny amplitude
    cols.append(3e4 * np.exp(-x / 5))                          # very fast decay
    noise = 20 * rng.standard_normal((E, 600))                 # pure-noise background (maxfev cases)
    y = np.concatenate([np.stack(cols, axis=1), noise], axis=1)
    p0 = (1.0, -1 / 30.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        popt, r2 = dosma.curve_fit(dosma.monoexponential, x, y, p0=p0)
        ier, nfev = scipy_info(x, y, p0)
        popt_none, r2_none = dosma.curve_fit(dosma.monoexponential, x, y)  # p0=None -> ones
        y3 = y.reshape(E, -1, 1, 1)
        tc_pf, r2_pf = dosma.MonoExponentialFit(tc0="polyfit", bounds=(0, np.inf),
                                                decimal_precision=3).fit(x, vols(y3))  # Cones recipe
        # the reference tests' far initial guess (test_fitting.py:98 p0=(1.0, 50.0), x = 1..4)
        x4 = np.asarray([1, 2, 3, 4])
        a = rng.random(50)
        b = rng.random(50)
        y4 = np.stack([dosma.monoexponential(x4, a[i], b[i]) for i in range(50)], axis=-1)
        popt_1_50, r2_1_50 = dosma.curve_fit(dosma.monoexponential, x4, y4, p0=(1.0, 50.0))
        popt_1_1, r2_1_1 = dosma.curve_fit(dosma.monoexponential, x4, y4)
        # integer input (DICOM-like int16)
        yi = np.clip(np.rint(900 * np.exp(-x[:, None] / rng.uniform(20, 70, 300))
                             + 15 * rng.standard_normal((E, 300))), -32768, 32767).astype(np.int16)
        popt_i, r2_i = dosma.curve_fit(dosma.monoexponential, x, yi, p0=p0)
        tc_i, r2_ti = dosma.MonoExponentialFit(tc0="polyfit", decimal_precision=3).fit(
            x, vols(yi.reshape(E, -1, 1, 1)))
    save("g3_edges.npz", x=x, y=y, popt=popt, r2=r2, ier=ier, nfev=nfev, popt_p0none=popt_none,
         r2_p0none=r2_none, tc_cones=tc_pf.A.reshape(-1), r2_cones=r2_pf.A.reshape(-1),
         x4=x4, y4=y4, popt_1_50=popt_1_50, r2_1_50=r2_1_50, popt_1_1=popt_1_1, r2_1_1=r2_1_1,
         y_int16=yi, popt_int16=popt_i, r2_int16=r2_i, tc_int16=tc_i.A.reshape(-1),
         r2_tc_int16=r2_ti.A.reshape(-1))


def g4():
    """Scan recipes: CubeQuant T1rho (cube_quant.py:170-176) with an ROI mask,
    Mapss echo subsets (mapss.py:170-204)."""
    rng = np.random.default_rng(4)
    shape = (24, 24, 6)
    tsl = np.asarray([1.0, 10.0, 30.0, 60.0])
    t1r = rng.uniform(20, 90, shape)
    s0 = rng.uniform(400, 1200, shape)
    y = s0 * np.exp(-tsl[:, None, None, None] / t1r) + 12 * rng.standard_normal((4,) + shape)
    y = np.rint(y).astype(np.int16)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    r = np.sqrt((zz - 12) ** 2 + (yy - 12) ** 2)
    mask = ((r > 6) & (r < 9)).astype(np.uint8)  # shell-like ROI
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tc, r2 = dosma.MonoExponentialFit(bounds=(0, 500), tc0="polyfit", decimal_precision=3,
                                          num_workers=NW).fit(tsl, vols(y), MV(mask, np.eye(4)))
        tc_nomask, r2_nomask = dosma.MonoExponentialFit(
            bounds=(0, 500), tc0="polyfit", decimal_precision=3, num_workers=NW).fit(tsl, vols(y))
        # Mapss: 7 echoes; T1rho from echoes 0-3, T2 from echoes [0,4,5,6]
        te_all = np.asarray([0.0, 10.0, 40.0, 80.0, 12.87, 25.69, 51.39])
        ym = np.stack([s0 * np.exp(-t / t1r) for t in te_all]) + 8 * rng.standard_normal((7,) + shape)
        ym = ym.astype(np.float32)
        idx = [0, 4, 5, 6]
        tc_t2, r2_t2 = dosma.MonoExponentialFit(bounds=(0, 100), tc0="polyfit", decimal_precision=3,
                                                num_workers=NW).fit(te_all[idx], vols(ym[idx]))
    save("g4_recipes.npz", tsl=tsl, y=y, mask=mask, tc=tc.A, r2=r2.A, tc_nomask=tc_nomask.A,
         r2_nomask=r2_nomask.A, te_mapss=te_all[idx], y_mapss=ym[idx], tc_mapss=tc_t2.A,
         r2_mapss=r2_t2.A)


def g5():
    """_process_params matrix (fitting.py:109-146; tests/core/test_fitting.py:325-412)."""
    rng = np.random.default_rng(5)
    shape = (10, 10, 4)
    x = np.asarray([0.5, 1.0, 2.0, 4.0])
    a = np.ones(shape)
    a[5:] = 1.5
    b = rng.random(shape) + 0.1
    b[:5] = 1.5
    y = np.stack([dosma.monoexponential(t, a, b) for t in x])
    y += 0.02 * rng.standard_normal(y.shape)
    out = {}
    ufunc = lambda v: 2 * np.abs(v) + 5  # noqa: E731
    cases = {
        "bounds_all": dict(out_bounds=(0, 1.2)),
        "bounds_second": dict(out_bounds=[(-np.inf, np.inf), (0, 1.2)]),
        "bounds_first": dict(out_bounds=[(0, 1.2)]),
        "nan_to_num": dict(out_bounds=(0, 1.2), nan_to_num=0.0),
        "ufunc_all": dict(out_ufuncs=ufunc),
        "ufunc_second": dict(out_ufuncs=[None, ufunc]),
        "ufunc_first": dict(out_ufuncs=[ufunc]),
        "r2_none": dict(r2_threshold=None),
        "r2_099": dict(r2_threshold=0.9999, nan_to_num=-1.0),
        "p0_tuple": dict(p0=(1.0, 0.5)),
    }
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k, kw in cases.items():
            popt, r2 = dosma.CurveFitter(dosma.monoexponential, **kw).fit(x, vols(y))
            out[f"popt_{k}"] = popt.A
            out[f"r2_{k}"] = r2.A
    save("g5_process_params.npz", x=x, y=y, **out)


def g6():
    """qDESS analytic T2 + RSS: the reference's QDess class on seeded echoes."""
    from dosma.scan_sequences.mri.qdess import QDess

    rng = np.random.default_rng(6)
    shape = (24, 20, 6)
    e1 = rng.uniform(20, 800, shape)
    e2 = e1 * rng.uniform(0.02, 0.9, shape)
    e1[0, 0, 0] = 0          # division by zero -> inf -> nan_to_num
    e1[1, 1, 1] = e2[1, 1, 1] = 0  # 0/0 -> nan
    e2[2, 2, 2] = 0          # log(0)
    out = {}
    pars = dict(gl_area=3132, tg=1904, tr=20.36, te=6.428, alpha=20.0, t1=1200.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for dt in (np.float32, np.float64, np.int16):
            a, b = e1.astype(dt), e2.astype(dt)
            q = QDess([MV(a, np.eye(4)), MV(b, np.eye(4))])
            tag = np.dtype(dt).name
            out[f"e1_{tag}"], out[f"e2_{tag}"] = a, b
            out[f"t2_{tag}"] = q.generate_t2_map(**pars).volumetric_map.A
            out[f"t2_sup_{tag}"] = q.generate_t2_map(suppress_fat=True, suppress_fluid=True, decimals=3,
                                                     nan_bounds=(0, 80), **pars).volumetric_map.A
            out[f"t2_raw_{tag}"] = q.generate_t2_map(nan_bounds=None, nan_to_num=None, decimals=None,
                                                     **pars).volumetric_map.A
            out[f"rss_{tag}"] = q.calc_rss().A
    save("g6_qdess.npz", pars=np.array([pars[k] for k in ("gl_area", "tg", "tr", "te", "alpha", "t1")]), **out)


def g7():
    """Bi-exponential model: dosma.curve_fit / CurveFitter with func=biexponential."""
    rng = np.random.default_rng(7)
    E = 12
    x = np.linspace(4.0, 92.0, E)
    n = 1500
    a1 = rng.uniform(300, 900, n)
    a2 = rng.uniform(200, 700, n)
    ts = rng.uniform(5, 15, n)     # short compartment
    tl = rng.uniform(40, 90, n)    # long compartment
    clean = a1 * np.exp(-x[:, None] / ts) + a2 * np.exp(-x[:, None] / tl)
    y = clean + rng.normal(0, 2.0, clean.shape)
    y[:, :20] = 0.0                                   # skip rule
    y[:, 20:60] = rng.normal(0, 5.0, (E, 40))         # pure noise: failures / garbage
    y[:, 60:80] = clean[:, 60:80]                     # noise-free
    p0 = (500.0, -0.1, 500.0, -0.02)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        popt, r2 = dosma.curve_fit(dosma.biexponential, x, y, p0=p0, num_workers=NW)
        ier, nfev = scipy_info(x, y, p0, func=dosma.biexponential)
        out.update(popt=popt, r2=r2, ier=ier, nfev=nfev)
        # default p0 (ones) -- mostly failures: the failure class must agree
        popt1, r21 = dosma.curve_fit(dosma.biexponential, x, y[:, :300], num_workers=NW)
        out.update(popt_ones=popt1, r2_ones=r21)
        # float32 / int16 samples, per-voxel p0 for two parameters, y_bounds
        y32 = y.astype(np.float32)
        p0v = {"a1": np.abs(y32[0].astype(np.float64)) * 0.6 + 1.0, "b1": -0.1, "a2": 400.0,
               "b2": np.full(n, -0.015)}
        popt2, r22 = dosma.curve_fit(dosma.biexponential, x, y32, p0=p0v, y_bounds=(-50, 1500), num_workers=NW)
        out.update(y32=y32, p0v_a1=p0v["a1"], p0v_b2=p0v["b2"], popt_f32=popt2, r2_f32=r22)
        # CurveFitter on volumes with a mask and host post-processing
        shape = (15, 10, 10)
        yv = y.reshape((E,) + shape)
        mask = (rng.uniform(size=shape) < 0.6)
        cf = dosma.CurveFitter(dosma.biexponential, p0=p0, out_ufuncs=[None, lambda v: 1 / np.abs(v), None,
                                                                     lambda v: 1 / np.abs(v)],
                               out_bounds=(0, 2000), r2_threshold=0.9, nan_to_num=0.0, num_workers=NW)
        pm, rm = cf.fit(x, vols(yv), mask=MV(mask.astype(np.uint8), np.eye(4)))
        out.update(mask=mask, popt_cf=pm.A, r2_cf=rm.A)
    save("g7_biexp.npz", x=x, y=y, p0=np.array(p0), **out)


def g8():
    """QuantitativeValue.to_metrics (quant_vals.py:145-229; pinned by tests/core/test_quant_vals.py:52-174): the real
    reference's DataFrame for label maps, ``labels`` subsets, ``bounds`` with all four ``closed`` modes, non-finite
    voxels, float64 and float32 maps."""
    from dosma.core.quant_vals import T2

    rng = np.random.default_rng(8)
    shape = (40, 36, 12)
    vol = rng.uniform(5.0, 95.0, shape)
    vol[rng.uniform(size=shape) < 0.02] = np.nan
    vol[rng.uniform(size=shape) < 0.01] = np.inf
    vol[rng.uniform(size=shape) < 0.01] = -np.inf
    vol.reshape(-1)[:600] = np.round(vol.reshape(-1)[:600])          # exact ties and values ON the bounds
    vol[3, 3, 3], vol[4, 4, 4], vol[5, 5, 5] = 20.0, 60.0, 60.0      # both interval ends present
    lab = rng.integers(0, 5, shape).astype(np.uint8)                 # labels 0 (background) .. 4
    lab[:, :4, :] = 0
    lab[vol == 20.0] = 2
    out = dict(vol=vol, labels=lab)

    def record(tag, df):
        out[f"{tag}_category"] = np.array(list(df["Category"]), dtype="U16")
        for col, key in (("Mean", "mean"), ("Std", "std"), ("Median", "median"), ("# Voxels", "count")):
            out[f"{tag}_{key}"] = np.asarray(df[col], dtype=np.float64)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for dt in (np.float64, np.float32):
            tag = np.dtype(dt).name
            qv = T2(MV(vol.astype(dt), np.eye(4)))
            mask = MV(lab, np.eye(4))
            record(f"{tag}_nomask", qv.to_metrics())
            record(f"{tag}_auto", qv.to_metrics(mask))
            record(f"{tag}_subset", qv.to_metrics(mask, labels={3: "tc", 1: "fc"}))
            for closed in ("right", "left", "both", "neither"):
                record(f"{tag}_b_{closed}", qv.to_metrics(mask, bounds=(20.0, 60.0), closed=closed))
            record(f"{tag}_b_nomask", qv.to_metrics(bounds=(20.0, 60.0)))
            # a label that selects nothing (all of its voxels out of bounds) -> NaN statistics, 0 voxels
            record(f"{tag}_empty", qv.to_metrics(mask, labels={4: "men"}, bounds=(1000.0, 2000.0)))
    save("g8_to_metrics.npz", **out)


def g9():
    """``generate_mask`` pre / post logic of every segmentation template (oaiunet2d.py:140-175, 291-320, 344-345;
    stanford_qdess.py:158-205; seg_model.py:114-127), run by the reference's OWN code.  Keras / TensorFlow and the weight
    files are absent here, so the instances are made with ``object.__new__`` and get a ``seg_model`` whose
    ``predict`` is a seeded per-pixel function of the array the reference hands it: sigmoid(a_c * v + b_c) per class.  The
    NETWORK is therefore not what this fixture pins -- the reformat to SAGITTAL, the preprocessing, the (slice, x, y, 1)
    layout, ``> sigmoid_threshold``, the class order, the per-class clones and the reformat back are.  Stored: the input
    volume + affine, the array ``predict`` received, the probabilities it returned, and every output volume / affine."""
    import dosma.models.oaiunet2d as ref_o
    import dosma.models.seg_model as ref_sm
    import dosma.models.stanford_qdess as ref_s

    for mod in (ref_o, ref_sm, ref_s):  # destructor of KerasSegModel: K.clear_session() (seg_model.py:105-106), Keras absent
        if not hasattr(mod, "K"):
            mod.K = type("K", (), {"clear_session": staticmethod(lambda: None)})

    class Predictor:
        def __init__(self, a, b):
            self.a, self.b = np.asarray(a, np.float64), np.asarray(b, np.float64)
            self.seen = None

        def predict(self, v, batch_size=None, verbose=0):
            assert v.ndim == 4 and v.shape[-1] == 1
            self.seen = np.array(v)
            z = v.astype(np.float64) * self.a + self.b          # (S, H, W, C) logits
            self.logits = z
            return (1.0 / (1.0 + np.exp(-z))).astype(np.float32)

    rng = np.random.default_rng(9)
    H, W, S = 64, 32, 3                                          # sagittal frame: (SI, AP, LR); a size the 6-level network accepts
    # a volume whose SAGITTAL view is (H, W, S), stored in four orientations with anisotropic spacing + an offset
    base = rng.gamma(2.0, 150.0, (H, W, S)).astype(np.float32)
    base[:3] = 0.0
    sag_aff = np.array([[0, 0, 1.5, -40.0], [0, -0.35, 0, 70.5], [-0.4, 0, 0, 33.25], [0, 0, 0, 1.0]])
    sag = MV(base, sag_aff)
    assert sag.orientation == ("SI", "AP", "LR"), sag.orientation
    orients = {"sag": ("SI", "AP", "LR"), "ax": ("AP", "LR", "SI"), "cor_flip": ("IS", "RL", "AP"), "perm": ("LR", "PA", "IS")}
    out = dict(orient_names=np.array(list(orients)), a4=np.array([0.9, -1.1, 0.7, 1.3]), b4=np.array([0.15, 0.4, -0.55, -0.2]))
    a4, b4 = out["a4"], out["b4"]
    raw_scale = 1.0 / 300.0                                       # un-whitened templates see raw intensities (0 .. ~2000)
    templates = {
        "iwoai": (ref_o.IWOAIOAIUnet2D, a4 * raw_scale, b4 - 0.8, None),
        "iwoai_norm": (ref_o.IWOAIOAIUnet2DNormalized, a4, b4, None),
        "oai": (ref_o.OAIUnet2D, a4[:1], b4[:1], None),
        "stanford": (ref_s.StanfordQDessUNet2D, a4, b4, None),
        "stanford_thr": (ref_s.StanfordQDessUNet2D, a4, b4, 0.7),
    }
    min_margin = np.inf
    for oname, orient in orients.items():
        vol_in = sag.reformat(orient)                            # a copy in the other orientation (med_volume.py:177-275)
        out[f"{oname}_vol"] = np.array(vol_in.volume)
        out[f"{oname}_affine"] = np.array(vol_in.affine)
        for tname, (cls, a, b, thr) in templates.items():
            model = object.__new__(cls)
            model.batch_size = 16
            model.seg_model = Predictor(a, b)
            if thr is not None:
                model.sigmoid_threshold = thr
            res = model.generate_mask(vol_in)
            tag = f"{oname}_{tname}"
            out[f"{tag}_net_in"] = model.seg_model.seen
            cut = 0.0 if thr is None else float(np.log(thr / (1 - thr)))
            min_margin = min(min_margin, float(np.abs(model.seg_model.logits - cut).min()))
            # pixels whose logit is within 2e-3 of the cut (S, H, W, C): a network that evaluates a_c * v + b_c in another
            # arithmetic (the GPU test's pass-through weights) may legitimately land on the other side there
            out[f"{tag}_near"] = np.abs(model.seg_model.logits - cut) < 2e-3
            if isinstance(res, dict):
                out[f"{tag}_keys"] = np.array(list(res.keys()))
                items = list(res.items())
            else:
                out[f"{tag}_keys"] = np.array([], dtype="U4")
                items = [("", res)]
            for k, m in items:
                assert m.volume.dtype == np.uint8 and m.orientation == tuple(orient)
                out[f"{tag}_mask_{k}"] = np.array(m.volume)
                out[f"{tag}_affine_{k}"] = np.array(m.affine)
    # 4D dual-echo input of the Stanford template (stanford_qdess.py:172-178): RSS of the two echoes first
    e = np.stack([base, rng.gamma(2.0, 90.0, (H, W, S)).astype(np.float32)], axis=-1)
    mv4 = MV(e, sag_aff)
    model = object.__new__(ref_s.StanfordQDessUNet2D)
    model.batch_size = 16
    model.seg_model = Predictor(a4, b4)
    res = model.generate_mask(mv4)
    out["dual_vol"] = e
    out["dual_net_in"] = model.seg_model.seen
    out["dual_near"] = np.abs(model.seg_model.logits) < 2e-3
    for k, m in res.items():
        out[f"dual_mask_{k}"] = np.array(m.volume)
    out["dual_keys"] = np.array(list(res.keys()))
    # whiten_volume on float32 and float64 volumes, both eps (seg_model.py:114-127; tests/models/test_oaiunet2d.py:61-81)
    for dt in (np.float32, np.float64):
        for eps in (0.0, 1e-8):
            out[f"whiten_{np.dtype(dt).name}_{eps:g}"] = ref_sm.whiten_volume(base.astype(dt), eps=eps)
    out["sag_vol"], out["sag_aff"] = base, sag_aff
    out["min_logit_margin"] = np.array(min_margin)
    print(f"  smallest |logit - cut| over all cases: {min_margin:.3e}")
    save("g9_generate_mask.npz", **out)


def g0():
    """More samples per voxel than any kernel keeps (65 > 64): MonoExponentialFit with a numeric tc0 and with
    tc0="polyfit", masked, and the plain CurveFitter (fitting.py:607-749, 238-458) -- the route ADVICE r5 found broken."""
    rng = np.random.default_rng(10)
    shape = (6, 5, 4)
    E = 65
    x = np.linspace(2.0, 90.0, E)
    s0 = rng.uniform(300, 1500, shape)
    t2 = rng.uniform(15, 80, shape)
    y = s0 * np.exp(-x.reshape(-1, 1, 1, 1) / t2) + 6.0 * rng.standard_normal((E,) + shape)
    y[:, 0, 0, 0] = 0          # skip rule
    y[3, 1, 1, 1] = 0          # a zero sample: eps enters the log of the polyfit guess
    mask = rng.random(shape) > 0.3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tc_def, r2_def = dosma.MonoExponentialFit(decimal_precision=6).fit(x, vols(y))
        tc_pf, r2_pf = dosma.MonoExponentialFit(tc0="polyfit", decimal_precision=3).fit(x, vols(y), mask)
        popt, r2 = dosma.CurveFitter(dosma.monoexponential, p0=(1.0, -1 / 30.0)).fit(x, vols(y), mask=mask)
    save("g0_many_samples.npz", x=x, y=y, mask=mask, tc_default=tc_def.A, r2_default=r2_def.A,
         tc_polyfit_masked=tc_pf.A, r2_polyfit_masked=r2_pf.A, popt_masked=popt.A, r2_masked=r2.A)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["g0", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9"]
    for name in which:
        t = time.time()
        print(name, "...")
        globals()[name]()
        print(f"  {time.time() - t:.1f}s")

