"""Parity-mode U-Net against the fp64 restatement (oracle/unet_oracle.py) over MANY slice shapes -- every square size that is a
multiple of 32 up to 512 and a set of non-square ones -- so that every kernel family / tiling choice the dispatch can make
(8 x 32 image tiles, 16-row tiles of conv_c4_kernel / deconv_d4_kernel, the flattened tiling, the general kernel, the dedicated
top-level kernels and their fallbacks) is exercised at the north_star tolerance (1e-3 abs on the logits), not only the sizes the
test-suite pins.  Test infrastructure (imports oracle/): run on the GPU box,

    python scripts/unet_size_sweep.py [--slices 2] [--quick]

prints one line per shape: max |dlogit|, the kernel families of the trace, and FAIL where the tolerance is missed."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from dosma_amd import _lib as L
from oracle import unet_oracle as uo
from test_unet_gpu import weights_in_abi_order

ap = argparse.ArgumentParser()
ap.add_argument("--slices", type=int, default=2)
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()

w = uo.make_weights(seed=11, bn="realistic")
tensors = weights_in_abi_order(w)
shapes = [(s, s) for s in range(32, 513, 32)]
shapes += [(384, 512), (512, 384), (96, 288), (288, 96), (160, 64), (64, 160), (224, 384), (352, 480), (32, 512), (512, 32), (416, 96)]
if args.quick:
    shapes = [(64, 64), (160, 96), (256, 256), (320, 384)]
bad = 0
t_all = time.time()
for (H, W) in shapes:
    rng = np.random.default_rng(H * 1000 + W)
    yy, xx = np.mgrid[0:H, 0:W]
    blob = np.exp(-(((yy - H / 2) / (H / 4)) ** 2 + ((xx - W / 2) / (W / 3)) ** 2))
    vol = (rng.standard_normal((args.slices, H, W)) * 60 + 250 * blob[None] + 80).astype(np.float32)
    xw = uo.whiten_volume(vol.astype(np.float64)).astype(np.float32)
    t0 = time.time()
    ref = uo.forward(w, xw, dtype="float64")
    t_ref = time.time() - t0
    eng = L.Unet2dEngine(tensors, H, W, max_batch=args.slices + 1, precision="fp16x3")
    logits, mask = eng.forward_host(vol, whiten=True, eps=0.0)
    err = float(np.abs(logits - ref).max())
    fams = sorted({"/".join(t.split(":", 1)[1].split("/")[:2]) for t in eng.trace() if ":" in t and not t.startswith(("pool", "head"))})
    # the mask may differ from the restatement's only inside the tolerance band
    band = np.abs(ref) < 1e-3
    flips = (mask.astype(bool) != (ref > 0)) & ~band
    ok = err < 1e-3 and not flips.any()
    bad += not ok
    eng.close()
    print(f"{H:3d} x {W:3d}  max |dlogit| {err:.2e} (span {np.abs(ref).max():6.1f})  {'ok  ' if ok else 'FAIL'}  oracle {t_ref:5.1f} s  {' '.join(fams)}", flush=True)
print(f"{len(shapes)} shapes, {bad} failures, {time.time() - t_all:.0f} s")
sys.exit(1 if bad else 0)
