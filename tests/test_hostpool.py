"""dosma_amd/_hostpool.py: large result arrays live in page-locked host memory and are recycled when the caller drops
them -- callers must keep owning their results (the reference hands out fresh numpy arrays, fitting.py:205-215)."""
import gc

import numpy as np
import pytest

from dosma_amd import _hostpool


def test_small_or_gpu_less_requests_are_plain_numpy():
    a = _hostpool.empty((10, 3), np.float64)
    assert a.shape == (10, 3) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
    from dosma_amd import _lib
    if _lib.load().qmri_device_count() <= 0:  # no GPU: no HIP host allocator either
        b = _hostpool.empty(1 << 20, np.float64)
        assert b.shape == (1 << 20,) and _hostpool.cached_bytes() == 0


@pytest.mark.gpu
def test_results_are_owned_by_the_caller_and_blocks_are_recycled():
    import dosma_amd as dm
    from dosma_amd import _lib as L

    _hostpool.trim()
    rng = np.random.default_rng(0)
    n = 400_000
    x = np.arange(1.0, 9.0) * 10
    t2 = rng.uniform(20, 70, n)
    y = (1000 * np.exp(-x[:, None] / t2) + rng.standard_normal((8, n)) * 5).astype(np.float32)
    o1 = L.monoexp_fit_host(x, y, p0=(1.0, -1 / 30.0))
    keep = {k: v.copy() for k, v in o1.items()}
    o2 = L.monoexp_fit_host(x, y[:, ::-1].copy(), p0=(1.0, -1 / 30.0))   # a second result while the first is alive
    for k in keep:  # the first result was not overwritten, the second is the mirrored one
        assert np.array_equal(o1[k], keep[k], equal_nan=True)
    assert np.array_equal(o2["r2"][::-1], o1["r2"])
    assert o1["popt"].flags.writeable and o1["popt"].base is not None
    view = o1["popt"][:, 1]          # a view keeps the block alive
    before = _hostpool.cached_bytes()
    del o1, o2
    gc.collect()
    mid = _hostpool.cached_bytes()
    assert mid > before              # r2 of the first result and the whole second result went back to the free list
    assert np.array_equal(view, keep["popt"][:, 1], equal_nan=True)
    del view
    gc.collect()
    assert _hostpool.cached_bytes() > mid
    o3 = L.monoexp_fit_host(x, y, p0=(1.0, -1 / 30.0))   # served from the free list
    assert _hostpool.cached_bytes() < mid + 8 * n * 2 and np.array_equal(o3["r2"], keep["r2"])
    del o3
    gc.collect()
    _hostpool.trim()
    assert _hostpool.cached_bytes() == 0
    # the drop-in API hands out such arrays too
    vols = [dm.MedicalVolume(v.reshape(100, 100, 40), np.eye(4)) for v in y]
    tc, r2 = dm.MonoExponentialFit(tc0=30.0).fit(x, vols)
    assert tc.volume.shape == (100, 100, 40) and np.isfinite(r2.volume).all()
