"""In-kernel cycle accounting of conv_s3_kernel (experiment build, QMRI_S3_DBG & 1024): per work item, wave 0 of every block:
main loop | epilogue (affine part) | tile switch.  Run with DOSMA_AMD_LIB=dosma_amd/libqmri_hip_exp.so QMRI_S3_DBG=1024[+bits]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L

lib = L.load()
lib.qmri_s3_debug_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
rng = np.random.default_rng(0)
SHAPES = [("d0c2 32->32 @384", 32, 32, 384, 16, False), ("u0c1 64->32 @384", 64, 32, 384, 16, False),
          ("d1c2 64->64 @192", 64, 64, 192, 64, False), ("u1c1 128->64 @192", 128, 64, 192, 32, False),
          ("d3c2 256->256 @48", 256, 256, 48, 160, False), ("u0dc 64->32 @192 T", 64, 32, 192, 32, True)]
print(f"QMRI_S3_DBG={os.environ.get('QMRI_S3_DBG')}")
for name, cin, cout, hw, b, tr in SHAPES:
    x = rng.standard_normal((b, hw, hw, cin)).astype(np.float32)
    k = (rng.standard_normal((3, 3, cout, cin) if tr else (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    L.conv2d_nhwc_host(x, k, bias, relu=True, transposed=tr, precision="fp16x3")
    lib.qmri_s3_debug_stats(out, 1)
    L.conv2d_nhwc_host(x, k, bias, relu=True, transposed=tr, precision="fp16x3")
    lib.qmri_s3_debug_stats(out, 1)
    n = max(out[3], 1)
    print(f"{name:22s} work items {out[3]:7d}  main {out[0]/n:9.0f}  epilogue {out[1]/n:8.0f} (affine {out[4]/n:6.0f})  switch {out[2]/n:7.0f}  cycles per work item")
