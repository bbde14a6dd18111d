"""Phase cycles of out0_kernel (experiment build, QMRI_ENC0=0 so that enc0_kernel does not write the same counters)."""
import ctypes, os, sys
os.environ["QMRI_ENC0"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L
from oracle import unet_oracle as uo  # weights generator only
from dosma_amd.models import weights as W

lib = L.load()
lib.qmri_enc0_debug_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
w = uo.make_weights(seed=3)
eng = L.Unet2dEngine(W.to_abi_order(w), 384, 384, n_classes=4, max_batch=32, precision="fp16x3")
x = np.random.default_rng(0).standard_normal((32, 384, 384)).astype(np.float32)
eng.forward_host(x, whiten=True, eps=0.0)
lib.qmri_enc0_debug_stats(out, 1)
eng.forward_host(x, whiten=True, eps=0.0)
lib.qmri_enc0_debug_stats(out, 1)
n = max(out[6], 1)
names = ["halo requests", "MFMA loop", "classifier + stores", "wait + barrier"]
print("tiles", out[6], " cycles per tile (wave 0): " + "  ".join(f"{nm} {out[i]/n:.0f}" for i, nm in enumerate(names)), " sum", sum(out[i] for i in range(4)) / n)
