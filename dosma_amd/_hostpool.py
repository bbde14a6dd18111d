"""Result arrays in page-locked host memory, recycled.

The reference returns freshly allocated numpy arrays (``dosma/core/fitting.py:205-215``: ``np.full`` + scatter).  On the
GPU path a fresh 0.7 GB result costs more than the fit: the operating system zero-fills every page on first touch
(65 ms per GB on the benchmark host) and the download has to go through a staging buffer.  Here large
results live in ``hipHostMalloc`` memory handed out by ``empty()``: the pages are resident and the copy is a true DMA.  A
block goes back to the free list when the last numpy view of it is garbage-collected (a ``weakref`` finalizer on the
exporting buffer object), so callers own their results exactly as before; nothing is ever overwritten under them.

``DOSMA_AMD_HOST_POOL_GB`` (default 8; 0 switches the pool off) bounds ALL page-locked bytes of the pool -- blocks on the
free list plus blocks still owned by live results (pinned memory cannot be swapped); beyond it ``empty()`` hands out plain
numpy arrays.  ``DOSMA_AMD_HOST_POOL=0`` switches the pool off as well.

Fork: ROCm maps page-locked allocations so that a ``fork()``ed child does not inherit them; the library re-enables
inheritance on every block (``qmri_host_alloc``: ``madvise(MADV_DOFORK)``), so a ``multiprocessing`` worker forked while
results are alive can still READ them (tests/test_hostpool.py::test_forked_child_reads_a_pooled_result); a child never
frees or recycles the parent's blocks.
"""
import atexit
import ctypes
import os
import threading
import weakref

import numpy as np

_MIN_BYTES = 1 << 20        # smaller arrays: plain numpy
_GRANULE = 2 << 20          # size classes of 2 MB
_lock = threading.RLock()   # re-entrant: a finalizer (_release) can run inside a garbage collection that an allocation under
                            # the lock triggers on the same thread
_free = {}                  # size -> [pointers]
_cached = 0                 # bytes on the free list
_live = 0                   # bytes handed out and not yet released
_closed = False
_pid = os.getpid()          # blocks belong to the process that allocated them (a forked worker never frees or reuses them)


def _cap():
    try:
        return int(float(os.environ.get("DOSMA_AMD_HOST_POOL_GB", "8")) * (1 << 30))
    except ValueError:
        return 8 << 30


def _release(ptr, size):
    global _cached, _live
    if _closed or os.getpid() != _pid:  # interpreter shutdown / forked child: leave the block alone
        return
    cap = _cap()
    with _lock:
        _live -= size
        if _cached + _live + size <= cap:
            lst = _free.get(size)
            if lst is None:
                lst = _free[size] = []
            lst.append(ptr)
            _cached += size
            return
    try:
        from dosma_amd import _lib
        _lib.load().qmri_host_free(ctypes.c_void_p(ptr))
    except Exception:       # pragma: no cover - nothing sensible to do in a finalizer
        pass


def _take(size):
    global _cached, _live
    with _lock:
        lst = _free.get(size)
        if lst:
            _cached -= size
            _live += size
            return lst.pop()
    return None


def empty(shape, dtype):
    """``numpy.empty(shape, dtype)`` in page-locked memory (plain numpy for small arrays, without a GPU, or when the
    allocation fails)."""
    global _live, _cached
    dt = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(int(s) for s in shape)
    count = 1
    for s in shape:
        count *= s
    nbytes = count * dt.itemsize
    cap = _cap()
    if (nbytes < _MIN_BYTES or _closed or cap <= 0 or os.getpid() != _pid
            or os.environ.get("DOSMA_AMD_HOST_POOL", "1") == "0"):
        return np.empty(shape, dt)
    size = (nbytes + _GRANULE - 1) // _GRANULE * _GRANULE
    ptr = _take(size)
    if ptr is None:
        with _lock:
            fits = _live + size <= cap
            drop = []
            if fits:
                _live += size  # reserved under the same lock as the check: concurrent callers cannot pass it together
                # make room among the cached blocks of other sizes first: the total stays below the cap
                for sz in sorted(_free, reverse=True):
                    while _free[sz] and _live + _cached > cap:
                        drop.append((sz, _free[sz].pop()))
                        _cached -= sz
        if not fits:
            return np.empty(shape, dt)
        try:
            from dosma_amd import _lib
            lib = _lib.load()
            out = ctypes.c_void_p()
            while drop:
                lib.qmri_host_free(ctypes.c_void_p(drop[-1][1]))
                drop.pop()
            if lib.qmri_device_count() > 0 and lib.qmri_host_alloc(size, ctypes.byref(out)) == 0 and out.value:
                ptr = out.value
        except Exception:
            ptr = None
        if ptr is None:
            with _lock:
                _live -= size  # the reservation is rolled back; blocks that were not freed go back on the free list
                for sz, p in drop:
                    _free.setdefault(sz, []).append(p)
                    _cached += sz
            return np.empty(shape, dt)
    buf = (ctypes.c_ubyte * size).from_address(ptr)
    fin = weakref.finalize(buf, _release, ptr, size)
    fin.atexit = False  # nothing to recycle at interpreter exit (and no HIP call from an exit handler or a forked child)
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def trim():
    """Return every cached block to the system (tests, long-running services between jobs)."""
    global _cached
    with _lock:
        old = dict(_free)
        _free.clear()
        _cached = 0
    blocks = [(p, s) for s, lst in old.items() for p in lst]
    if blocks:
        from dosma_amd import _lib
        lib = _lib.load()
        for p, _ in blocks:
            lib.qmri_host_free(ctypes.c_void_p(p))


def cached_bytes():
    return _cached


def live_bytes():
    """Page-locked bytes currently owned by results that are still alive."""
    return _live


@atexit.register
def _shutdown():
    global _closed
    _closed = True
