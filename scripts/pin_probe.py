"""Probe: cost of hipHostRegister on a numpy volume vs pageable / pinned copy rates (host-entry design input)."""
import ctypes, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L
L.load()
import torch  # noqa: F401  (same HIP runtime)
hip = ctypes.CDLL(None)
for name in ("hipHostRegister", "hipHostUnregister", "hipMalloc", "hipMemcpy", "hipFree", "hipDeviceSynchronize", "hipMemcpyAsync"):
    getattr(hip, name).restype = ctypes.c_int
n = 1342177280  # bytes: 8 x 41.9M x f32
a = np.ones(n // 4, np.float32)
out = np.empty(n // 4, np.float32)
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(n)) == 0
def t(f, label, nbytes=n):
    hip.hipDeviceSynchronize(); t0 = time.perf_counter(); f(); hip.hipDeviceSynchronize(); dt = time.perf_counter() - t0
    print(f"{label:40s} {dt*1e3:8.1f} ms  {nbytes/dt/1e9:6.1f} GB/s")
H2D, D2H = 1, 2
t(lambda: hip.hipMemcpy(d, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), H2D), "pageable H2D (first touch)")
t(lambda: hip.hipMemcpy(d, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), H2D), "pageable H2D")
t(lambda: hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), d, ctypes.c_size_t(n), D2H), "pageable D2H (first touch of dst)")
t(lambda: hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), d, ctypes.c_size_t(n), D2H), "pageable D2H")
t(lambda: hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), 0), "hipHostRegister src")
t(lambda: hip.hipHostRegister(ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(n), 0), "hipHostRegister dst")
t(lambda: hip.hipMemcpy(d, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), H2D), "registered H2D")
t(lambda: hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), d, ctypes.c_size_t(n), D2H), "registered D2H")
t(lambda: hip.hipHostUnregister(ctypes.c_void_p(a.ctypes.data)), "hipHostUnregister src")
t(lambda: hip.hipHostUnregister(ctypes.c_void_p(out.ctypes.data)), "hipHostUnregister dst")
# CPU memcpy rate into a pinned staging buffer, 1 thread
pin = torch.empty(n // 4, dtype=torch.float32).pin_memory()
pv = pin.numpy()
t0 = time.perf_counter(); pv[:] = a; dt = time.perf_counter() - t0
print(f"{'numpy memcpy pageable->pinned, 1 thread':40s} {dt*1e3:8.1f} ms  {n/dt/1e9:6.1f} GB/s")
t(lambda: hip.hipMemcpy(d, ctypes.c_void_p(pv.ctypes.data), ctypes.c_size_t(n), H2D), "pinned H2D")
