"""QuantitativeValue.to_metrics on the GPU (csrc/region_stats.hip) against the host evaluation, which
is the reference's own numpy route (quant_vals.py:145-229: nanmean / nanstd / nanmedian / count per label, "total").
Counts and medians must be exact (radix selection); means / standard deviations are fp64 sums in a different order
than numpy's pairwise summation -> 1e-12 relative."""
import numpy as np
import pytest

import dosma_amd as dm
from dosma_amd import _lib as L
from dosma_amd.quant_vals import T2, QuantitativeValue

pytestmark = pytest.mark.gpu


def _host_frame(qv, **kw):
    """The same call forced through the host route (a user callable makes to_metrics evaluate on the host)."""
    df = qv.to_metrics(fns={"_n": lambda v: v.size}, **kw)
    return df.drop(columns=["_n"])


def _check(gpu, host, f32=False):
    """float32 maps: numpy sums them in float32 (1e-7 relative at best), the kernel in float64 -> compared at 1e-5."""
    assert list(gpu["Category"]) == list(host["Category"])
    assert list(gpu["# Voxels"]) == list(host["# Voxels"])
    for col, tol in (("Median", 0.0), ("Mean", 1e-5 if f32 else 1e-12), ("Std", 1e-5 if f32 else 1e-11)):
        g, h = gpu[col].to_numpy(float), host[col].to_numpy(float)
        assert np.array_equal(np.isnan(g), np.isnan(h)), col
        ok = ~np.isnan(h)
        assert np.all(np.abs(g[ok] - h[ok]) <= tol * np.maximum(1.0, np.abs(h[ok]))), (col, g, h)


def _map(shape, dtype, seed, decimals=None):
    rng = np.random.default_rng(seed)
    v = rng.uniform(-20, 120, shape)
    if decimals is not None:
        v = np.around(v, decimals)  # many ties, as a rounded T2 map has
    v[rng.uniform(size=shape) < 0.05] = np.nan
    v[rng.uniform(size=shape) < 0.01] = np.inf
    v[rng.uniform(size=shape) < 0.01] = -np.inf
    return v.astype(dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("decimals", [None, 1])
def test_labelled_regions_match_numpy(dtype, decimals):
    shape = (64, 48, 20)
    v = _map(shape, dtype, 1, decimals)
    rng = np.random.default_rng(2)
    lab = rng.integers(0, 6, shape).astype(np.uint8)
    lab[lab == 4] = 0  # label 4 is asked for but empty -> NaN statistics, count 0
    qv = T2(dm.MedicalVolume(v, np.eye(4)))
    mask = dm.MedicalVolume(lab, np.eye(4))
    labels = {1: "fc", 2: "tc", 3: "pc", 4: "empty", 5: "men"}
    f32 = dtype == np.float32
    for kw in (dict(), dict(bounds=(0, 100)), dict(bounds=(0, 100), closed="both"), dict(bounds=(10.0, 10.0), closed="both"),
               dict(bounds=(0, 80), closed="neither"), dict(bounds=(0, 80), closed="left")):
        _check(qv.to_metrics(mask, labels, **kw), _host_frame(qv, mask=mask, labels=labels, **kw), f32)
    # labels taken from the mask, float label map, more labels than one kernel call takes (chunks of 15)
    _check(qv.to_metrics(mask), _host_frame(qv, mask=mask), f32)
    many = rng.integers(0, 40, shape).astype(np.float32)
    maskf = dm.MedicalVolume(many, np.eye(4))
    _check(qv.to_metrics(maskf, bounds=(0, 100)), _host_frame(qv, mask=maskf, bounds=(0, 100)), f32)


def test_whole_map_and_small_counts():
    v = _map((40, 40, 10), np.float64, 3)
    qv = T2(dm.MedicalVolume(v, np.eye(4)))
    _check(qv.to_metrics(), _host_frame(qv))
    _check(qv.to_metrics(bounds=(0, 100)), _host_frame(qv, bounds=(0, 100)))
    # regions of 0, 1, 2, 3 voxels: the even / odd median rule on the smallest cases
    lab = np.zeros(v.shape, np.int16)
    flat = lab.reshape(-1)
    finite = np.flatnonzero(np.isfinite(v.reshape(-1)))
    flat[finite[0]] = 1
    flat[finite[1:3]] = 2
    flat[finite[3:6]] = 3
    mask = dm.MedicalVolume(lab, np.eye(4))
    labels = {1: "one", 2: "two", 3: "three", 7: "none"}
    _check(qv.to_metrics(mask, labels), _host_frame(qv, mask=mask, labels=labels))


def test_raw_entry_and_errors():
    rng = np.random.default_rng(5)
    v = rng.standard_normal(100_003).astype(np.float32) * 1e3
    out = L.region_stats_host(v)
    assert out.shape == (1, 4) and out[0, 0] == v.size
    assert out[0, 3] == np.median(v) and abs(out[0, 1] - v.astype(np.float64).mean()) < 1e-9
    lab = rng.integers(0, 3, v.size)
    with pytest.raises(ValueError):
        L.region_stats_host(v, lab, keys=range(1, 20))
    with pytest.raises(ValueError):
        L.region_stats_host(v, lab[:10], keys=(1,))
    empty = L.region_stats_host(np.zeros(0, np.float64))
    assert empty[0, 0] == 0 and np.isnan(empty[0, 1:]).all()


def test_full_size_map_against_a_histogram():
    """512 x 512 x 160 map of one-decimal values with a 4-label mask (BASELINE's volume size).  The expectation comes
    from one np.bincount over (label, value code) -- an independent and cheap route to the exact counts / medians and
    to the moments -- instead of numpy's per-label nanmedian over 42 M voxels."""
    rng = np.random.default_rng(7)
    shape = (512, 512, 160)
    code = rng.integers(0, 1001, shape, dtype=np.int32)   # value = code / 10 in [0, 100]
    code[::7, ::5] = 0
    v = code / 10.0
    lab = (rng.integers(0, 50, shape) // 10 % 5).astype(np.uint8)  # labels 0..4
    qv = T2(dm.MedicalVolume(v, np.eye(4)))
    mask = dm.MedicalVolume(lab, np.eye(4))
    labels = {1: "fc", 2: "tc", 3: "pc", 4: "men"}
    gpu = qv.to_metrics(mask, labels, bounds=(0, 100))     # (0, 100]: value 0 is outside
    hist = np.bincount((lab.astype(np.int64) * 1001 + code).reshape(-1), minlength=5 * 1001).reshape(5, 1001)
    hist[:, 0] = 0
    vals = np.arange(1001) / 10.0
    rows = [hist[k] for k in (1, 2, 3, 4)] + [hist[1:].sum(axis=0)]
    for (_, g), h in zip(gpu.iterrows(), rows):
        n = int(h.sum())
        c = np.cumsum(h)
        lo, hi = vals[np.searchsorted(c, (n - 1) // 2 + 1)], vals[np.searchsorted(c, n // 2 + 1)]
        mean = float((h * vals).sum() / n)
        std = float(np.sqrt((h * (vals - mean) ** 2).sum() / n))
        assert g["# Voxels"] == n and g["Median"] == 0.5 * (lo + hi)
        assert abs(g["Mean"] - mean) < 1e-11 * mean and abs(g["Std"] - std) < 1e-10 * std
    # the same bits every time: the moments are per-block partial sums added in a fixed order (round 5; floating-point atomics
    # before that made the last bits of mean / std depend on the order the blocks finished in)
    for _ in range(3):
        again = qv.to_metrics(mask, labels, bounds=(0, 100))
        for col in ("Mean", "Std", "Median", "# Voxels"):
            assert np.array_equal(again[col].to_numpy(), gpu[col].to_numpy()), col


def test_device_pointer_entry_matches_the_host_entry():
    import torch

    rng = np.random.default_rng(9)
    n = 300_001
    v = rng.uniform(-5, 105, n)
    v[::11] = np.nan
    lab = rng.integers(0, 4, n).astype(np.uint8)
    want = L.region_stats_host(v, lab, keys=(1, 2, 3), bounds=(0, 100), closed="both")
    dev = torch.device("cuda", 0)
    tv, tl = torch.from_numpy(v).to(dev), torch.from_numpy(lab).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    got = L.region_stats_device(tv.data_ptr(), np.float64, n, tl.data_ptr(), np.uint8, keys=(1, 2, 3), bounds=(0, 100),
                                closed="both", stream=st)
    assert np.array_equal(got[:, [0, 3]], want[:, [0, 3]]) and np.allclose(got, want, rtol=1e-12, equal_nan=True)
    whole = L.region_stats_device(tv.data_ptr(), np.float64, n, stream=st)
    assert whole[0, 0] == np.isfinite(v).sum() and whole[0, 3] == np.nanmedian(v)


@pytest.mark.parametrize("tag", ["float64", "float32"])
def test_to_metrics_vs_reference_golden(golden, tag):
    """region_stats.hip against the REAL reference's DataFrames (tests/golden/g8_to_metrics.npz, produced by
    oracle/make_golden.py g8 from /root/reference/dosma/core/quant_vals.py:145-229): label maps, `labels` subsets,
    `bounds` with the four `closed=` modes (values exactly on both interval ends are in the map), NaN / +-inf voxels,
    an empty region, float64 and float32 maps.  Counts and medians exact; means / standard deviations to 1e-12
    (float32 maps: the reference sums in float32)."""
    from test_host_logic import G8_CALLS, check_against_g8

    g = golden("g8_to_metrics.npz")
    qv = T2(dm.MedicalVolume(g["vol"].astype(tag), np.eye(4)))
    mask = dm.MedicalVolume(g["labels"], np.eye(4))
    assert QuantitativeValue._region_stats_gpu(qv.volumetric_map.volume, None, None, None, "right") is not None  # the GPU route is live
    for case, call in G8_CALLS.items():
        check_against_g8(g, tag, case, call(qv, mask), exact_moments=False)
