"""Re-entrancy of the C ABI ("calls on different devices/streams may run concurrently from
different host threads"): several host threads drive the fit and the UNet engine at the same time through ctypes
(which releases the GIL for the duration of the call), each on its own inputs; every result must be bit-identical
to the same call made alone.  Covers the per-device cached slab buffers / downloader threads of
``qmri_monoexp_fit_host`` (qmri_capi.hip) and the thread-local ``qmri_last_error``.
"""
import threading

import numpy as np
import pytest

from dosma_amd import _lib as L
from oracle import unet_oracle as uo

pytestmark = pytest.mark.gpu


def _volume(seed, n):
    rng = np.random.default_rng(seed)
    x = np.arange(1, 9) * 10.0
    s0 = rng.uniform(300, 1500, n)
    t2 = rng.uniform(15, 80, n)
    y = s0 * np.exp(-x[:, None] / t2) + 18.0 * rng.standard_normal((8, n))
    y[:, rng.uniform(size=n) < 0.3] = 0.0
    return x, y.astype(np.float32)


POST = dict(inv_abs_b=True, bounds=((-np.inf, np.inf), (0, 100.0)), r2_threshold=0.9, nan_to_num=0.0, decimals=3)


def _same(a, b):
    return all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)


def test_fit_calls_from_many_host_threads():
    # sizes straddle the slab size of the host pipeline so that both the one-slab and the pipelined path run
    jobs = [(s, n) for s, n in zip(range(6), (50_000, 3_000_000, 777, 1_200_000, 65_536, 2_500_001))]
    inputs = [_volume(s, n) for s, n in jobs]
    kw = dict(init=L.INIT_LOGLIN, post=POST, want_tc=True, want_info=True)
    alone = [L.monoexp_fit_host(x, y, **kw) for x, y in inputs]
    for _ in range(2):
        got, errs = [None] * len(inputs), []

        def run(i):
            try:
                got[i] = L.monoexp_fit_host(*inputs[i], **kw)
            except Exception as e:  # noqa: BLE001 - reported below
                errs.append((i, e))

        ts = [threading.Thread(target=run, args=(i,)) for i in range(len(inputs))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        for i in range(len(inputs)):
            assert _same(got[i], alone[i]), f"job {i} differs when run concurrently"


def test_error_state_is_per_thread():
    x, y = _volume(1, 1000)
    seen = {}

    def bad():
        try:
            L.monoexp_fit_host(x[:5], y)  # x length mismatch -> ValueError before / inside the ABI
        except ValueError as e:
            seen["bad"] = str(e)

    def good():
        seen["good"] = L.monoexp_fit_host(x, y, init=L.INIT_LOGLIN, post=POST, want_tc=True)

    ts = [threading.Thread(target=bad), threading.Thread(target=good)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert "bad" in seen and np.isfinite(seen["good"]["r2"]).all()


def test_unet_engines_and_fit_concurrently():
    from test_unet_gpu import weights_in_abi_order
    tensors = weights_in_abi_order(uo.make_weights(seed=3))
    rng = np.random.default_rng(0)
    vols = [rng.uniform(0, 1000, (6, 32, 32)).astype(np.float32) for _ in range(3)]
    engines = [L.Unet2dEngine(tensors, 32, 32, max_batch=4) for _ in vols]
    alone = [e.forward_host(v, whiten=True, eps=1e-8) for e, v in zip(engines, vols)]
    x, y = _volume(9, 400_000)
    fit_alone = L.monoexp_fit_host(x, y, init=L.INIT_LOGLIN, post=POST, want_tc=True)
    got = [None] * len(vols)
    fit_got = {}

    def seg(i):
        for _ in range(5):
            got[i] = engines[i].forward_host(vols[i], whiten=True, eps=1e-8)

    def fit():
        for _ in range(3):
            fit_got["r"] = L.monoexp_fit_host(x, y, init=L.INIT_LOGLIN, post=POST, want_tc=True)

    ts = [threading.Thread(target=seg, args=(i,)) for i in range(len(vols))] + [threading.Thread(target=fit)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(len(vols)):
        assert np.array_equal(got[i][0], alone[i][0]) and np.array_equal(got[i][1], alone[i][1])
    assert _same(fit_got["r"], fit_alone)
    [e.close() for e in engines]
