"""SURVEY.md 8(f) rows N1 / N2 on the GPU: scan recipes (CubeQuant / Cones / Mapss) and the qDESS analytic
T2 map + RSS, against golden vectors produced by the real reference (g4, g6) and the numpy restatement."""
import numpy as np
import pytest

from dosma_amd import MedicalVolume
from dosma_amd.scan_sequences import Cones, CubeQuant, Mapss, QDess
from oracle import fit_oracle as fo

pytestmark = pytest.mark.gpu


def mvs(arr4):
    return [MedicalVolume(np.array(v), np.eye(4)) for v in arr4]


@pytest.mark.parametrize("tag", ["float32", "float64", "int16"])
def test_qdess_t2_vs_reference_golden(golden, tag):
    g = golden("g6_qdess.npz")
    gl, tg, tr, te, al, t1 = g["pars"]
    pars = dict(gl_area=gl, tg=tg, tr=tr, te=te, alpha=al, t1=t1)
    q = QDess(mvs([g[f"e1_{tag}"], g[f"e2_{tag}"]]))
    t2 = q.generate_t2_map(**pars)
    assert t2.NAME == "t2" and t2.volumetric_map.dtype == np.float64
    a, ref = t2.volumetric_map.volume, g[f"t2_{tag}"]
    # elementwise fp64: identical up to the last ulp of log(); after rounding to 1 decimal, equal except
    # where a value sits on a rounding boundary
    assert (a != ref).mean() < 1e-3 and np.abs(a - ref).max() <= 0.1 + 1e-12
    s = q.generate_t2_map(suppress_fat=True, suppress_fluid=True, decimals=3, nan_bounds=(0, 80), **pars)
    ref = g[f"t2_sup_{tag}"]
    assert (s.volumetric_map.volume != ref).mean() < 2e-3
    raw = q.generate_t2_map(nan_bounds=None, nan_to_num=None, decimals=None, **pars).volumetric_map.volume
    ref = g[f"t2_raw_{tag}"]
    assert np.allclose(raw, ref, rtol=1e-13, atol=1e-300, equal_nan=True)
    # the three pathological voxels: 1/0 -> DBL_MAX ratio, 0/0 -> 0 ratio, log(0)
    for idx in ((0, 0, 0), (1, 1, 1), (2, 2, 2)):
        assert a[idx] == g[f"t2_{tag}"][idx]
    rss = q.calc_rss()
    assert rss.dtype == np.float64 and np.allclose(rss.volume, g[f"rss_{tag}"], rtol=1e-15)
    with pytest.raises(ValueError):
        q.generate_t2_map(tr=tr, te=te, alpha=al, t1=t1)  # gl_area / tg missing (reference: ValueError)


def test_qdess_t2_large_volume_properties():
    """Full-size (384 x 384 x 160) through size-independent properties: the numpy restatement on a slab,
    scale invariance of the echo ratio, idempotent rounding."""
    rng = np.random.default_rng(0)
    shape = (384, 384, 160)
    e1 = rng.uniform(20, 800, shape).astype(np.float32)
    e2 = (e1 * rng.uniform(0.02, 0.9, shape).astype(np.float32)).astype(np.float32)
    pars = dict(gl_area=3132, tg=1904, tr=20.36, te=6.428, alpha=20.0, t1=1200.0)
    t2 = QDess(mvs([e1, e2])).generate_t2_map(**pars).volumetric_map.volume
    ref = fo.dess_t2_numpy(e1[:, :, :4], e2[:, :, :4], pars["tr"], pars["te"], pars["tg"], pars["alpha"],
                           pars["gl_area"], pars["t1"])
    assert (t2[:, :, :4] != ref).mean() < 1e-3
    t2s = QDess(mvs([e1 * 2, e2 * 2])).generate_t2_map(**pars).volumetric_map.volume  # exact in binary fp
    assert np.array_equal(t2, t2s)
    assert np.array_equal(np.around(t2, 1), t2) and t2.min() >= 0 and t2.max() <= 100


def test_scan_recipes_vs_reference_golden(golden):
    g = golden("g4_recipes.npz")
    cq = CubeQuant(mvs(g["y"]), g["tsl"])
    qv = cq.generate_t1_rho_map(mask=MedicalVolume(g["mask"], np.eye(4)))
    assert qv.NAME == "t1_rho" and "r2" in qv.additional_volumes
    tc = qv.volumetric_map.volume
    assert (tc != g["tc"]).mean() < 2e-3 and np.abs(tc - g["tc"]).max() <= 1e-3 + 1e-9
    assert np.allclose(qv.additional_volumes["r2"].volume, g["r2"], atol=1e-4)
    tc = cq.generate_t1_rho_map().volumetric_map.volume
    assert (tc != g["tc_nomask"]).mean() < 2e-3
    # Mapss: 7 volumes, T2 from echoes [0, 4, 5, 6] -- feed the 4 golden echoes into those slots
    te = np.zeros(7)
    vols7 = [None] * 7
    for slot, i in zip([0, 4, 5, 6], range(4)):
        te[slot] = g["te_mapss"][i]
        vols7[slot] = MedicalVolume(g["y_mapss"][i], np.eye(4))
    for slot in (1, 2, 3):
        te[slot] = 10.0 * slot
        vols7[slot] = vols7[0]
    t2 = Mapss(vols7, te).generate_t2_map()
    assert t2.NAME == "t2" and (t2.volumetric_map.volume != g["tc_mapss"]).mean() < 2e-3
    # Cones: upper bound inf lets every positive tc through (cones.py:21-27)
    g3 = golden("g3_edges.npz")
    y = g3["y"].reshape(8, -1, 1, 1)
    t2s = Cones(mvs(y), g3["x"]).generate_t2_star_map()
    assert t2s.NAME == "t2_star"
    assert (t2s.volumetric_map.volume.reshape(-1) != g3["tc_cones"]).mean() < 0.02
    # times given out of order are sorted like mapss.py:217-223
    order = [2, 0, 3, 1]
    cq2 = CubeQuant([mvs(g["y"])[i] for i in order], g["tsl"][order])
    assert np.array_equal(cq2.generate_t1_rho_map().volumetric_map.volume,
                          cq.generate_t1_rho_map().volumetric_map.volume)


def test_baseline_config0_size_vs_scipy():
    """BASELINE.json configs[0]: MonoExponentialFit on a synthetic 64 x 64 x 16 volume, 4 echoes -- the whole
    volume against the reference's own call (one scipy.optimize.curve_fit per voxel + its post-processing, restated in
    oracle/fit_oracle.py and pinned to the real reference by tests/test_oracle.py)."""
    import dosma_amd as dm
    from oracle import fit_oracle as fo

    rng = np.random.default_rng(64)
    shape, x = (64, 64, 16), np.array([10.0, 25.0, 45.0, 70.0])
    t2 = rng.uniform(15, 90, shape)
    s0 = rng.uniform(300, 1500, shape)
    vols = [(s0 * np.exp(-t / t2) + rng.normal(0, 4.0, shape)).astype(np.float32) for t in x]
    for v in vols:
        v[:8] = 0  # background slab: skip rule
    ys = [dm.MedicalVolume(v, np.eye(4)) for v in vols]
    tc, r2 = dm.MonoExponentialFit(bounds=(0, 100), tc0=30.0, decimal_precision=3).fit(x, ys)
    y2 = np.stack([v.reshape(-1) for v in vols])
    ref_tc, ref_r2, _ = fo.monoexp_fit_arrays(x, y2, tc0=30.0, bounds=(0, 100), decimal_precision=3, threads=8)
    ref_tc, ref_r2 = ref_tc.reshape(shape), ref_r2.reshape(shape)
    assert tc.shape == shape and tc.dtype == np.float64
    assert (tc.A[:8] == 0).all() and (r2.A[:8] == 0).all()
    differ = np.abs(tc.A - ref_tc) > 1e-4 * np.maximum(np.abs(ref_tc), 1.0)
    assert differ.mean() < 2e-3, differ.mean()  # voxels on a rounding / threshold boundary
    assert np.abs(r2.A - ref_r2)[~differ].max() < 1e-4


def test_baseline_config2_size_cubequant_roi():
    """BASELINE.json configs[2]: T1rho (CubeQuant), 4 spin-lock times, 384 x 384 x 120 int16 volumes, cartilage-mask
    ROI only (~2 % of the voxels).  Outside the ROI: the scatter fill; inside: the C restatement of the reference's
    scipy loop on every ROI voxel, and exact recovery of the noise-free truth."""
    import dosma_amd as dm
    from dosma_amd.scan_sequences import CubeQuant
    from oracle import fit_oracle as fo

    rng = np.random.default_rng(384)
    shape = (384, 384, 120)
    tsl = np.array([1.0, 10.0, 30.0, 60.0])
    t1r = rng.uniform(25, 70, shape)
    s0 = rng.uniform(800, 3000, shape)
    vols = [np.rint(s0 * np.exp(-t / t1r)).astype(np.int16) for t in tsl]
    mask = np.zeros(shape, np.uint8)
    mask[150:230, 120:260, 40:70] = (rng.uniform(size=(80, 140, 30)) < 0.85)  # a cartilage-plate-like slab
    assert 0.01 < mask.mean() < 0.03
    scan = CubeQuant([dm.MedicalVolume(v, np.eye(4)) for v in vols], tsl)
    qv = scan.generate_t1_rho_map(mask=dm.MedicalVolume(mask, np.eye(4)))
    t1map, r2 = qv.volumetric_map.A, qv.additional_volumes["r2"].A
    assert t1map.shape == shape and (t1map[mask == 0] == 0).all() and (r2[mask == 0] == 0).all()
    sel = mask.reshape(-1) > 0
    y2 = np.stack([v.reshape(-1)[sel] for v in vols]).astype(np.float64)
    ref_tc, ref_r2, _ = fo.monoexp_fit_arrays(tsl, y2, tc0="polyfit", bounds=(0, 500), decimal_precision=3, threads=8)
    got = t1map.reshape(-1)[sel]
    differ = np.abs(got - ref_tc) > 1e-4 * np.maximum(np.abs(ref_tc), 1.0)
    assert differ.mean() < 2e-3, differ.mean()
    ok = got > 0
    assert ok.mean() > 0.99 and np.abs(got[ok] - t1r.reshape(-1)[sel][ok]).max() < 0.5  # int16-rounded samples
