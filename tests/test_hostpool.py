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


@pytest.mark.gpu
def test_forked_child_reads_a_pooled_result():
    """page-locked allocations are MADV_DONTFORK under ROCm; the library switches inheritance back on
    (qmri_host_alloc), so a worker forked while a result is alive -- the reference's own multiprocessing pattern -- reads
    the parent's values instead of faulting, and never recycles the parent's blocks."""
    import os

    a = _hostpool.empty((4 << 20,), np.float64)     # 32 MB: from the pool
    if a.base is None:
        pytest.skip("pool is off")
    a[:] = np.arange(a.size)
    expect = float(a[::4097].sum())
    live = _hostpool.live_bytes()
    pid = os.fork()
    if pid == 0:  # child: touch the inherited pages, drop the view (must not free the parent's block), report
        code = 3
        try:
            ok = float(a[::4097].sum()) == expect and a[-1] == a.size - 1
            del a
            gc.collect()
            code = 0 if ok else 2
        finally:
            os._exit(code)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, f"child status {status:#x} (signal {status & 0x7f})"
    assert float(a[::4097].sum()) == expect and _hostpool.live_bytes() == live   # the parent still owns its block
    del a
    gc.collect()
    assert _hostpool.live_bytes() == live - (32 << 20)


@pytest.mark.gpu
def test_pinned_bytes_are_capped_including_live_results(monkeypatch):
    """DOSMA_AMD_HOST_POOL_GB bounds live + cached page-locked bytes; beyond it results are plain numpy arrays."""
    _hostpool.trim()
    gc.collect()
    base = _hostpool.live_bytes()
    monkeypatch.setenv("DOSMA_AMD_HOST_POOL_GB", repr((base + (48 << 20)) / (1 << 30)))   # 48 MB above what is alive now
    a = _hostpool.empty((2 << 20,), np.float64)    # 16 MB
    b = _hostpool.empty((2 << 20,), np.float64)
    assert a.base is not None and b.base is not None and _hostpool.live_bytes() == base + (32 << 20)
    c = _hostpool.empty((4 << 20,), np.float64)    # 32 MB more would exceed the cap: plain numpy
    assert c.base is None and _hostpool.live_bytes() == base + (32 << 20)
    del a, b
    gc.collect()
    assert _hostpool.live_bytes() == base and _hostpool.cached_bytes() <= (48 << 20)
    d = _hostpool.empty((5 << 20,), np.float64)    # 40 MB: fits once cached blocks of other sizes are dropped
    assert d.base is not None and _hostpool.live_bytes() + _hostpool.cached_bytes() <= base + (48 << 20)
    del c, d
    gc.collect()
    _hostpool.trim()
