"""The scenario of tests/test_unet_fullsize_gpu.py::test_512_logits_at_cfg5_batch, repeated: realistic-BatchNorm weights, one 160-slice
512 x 512 volume through a max_batch-160 engine and a max_batch-64 engine; per run the activation exponent the forward ended at (trace)
and where the logits differ from the first run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dosma_amd import _lib as L
from oracle import unet_oracle as uo
from test_unet_gpu import weights_in_abi_order

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
H = W = int(os.environ.get("BITS_HW", "512")); S = 160
w = uo.make_weights(seed=11, bn="realistic")
tensors = weights_in_abi_order(w)
rng = np.random.default_rng(5120)
yy, xx = np.mgrid[0:H, 0:W]
blob = np.exp(-(((yy - H / 2) / (H / 4)) ** 2 + ((xx - W / 2) / (W / 3)) ** 2))
vol = (rng.standard_normal((S, H, W)) * 60 + 250 * blob[None] + 80).astype(np.float32)
ref = None
for r in range(reps):
    mb = (160, 64)[r % 2]
    eng = L.Unet2dEngine(tensors, H, W, max_batch=mb, precision="fp16x3")
    logits, _ = eng.forward_host(vol, whiten=True, eps=0.0)
    shifts = [t for t in eng.trace() if t.startswith("act_shift")]
    eng.close()
    if ref is None:
        ref = logits; print(f"run 0 (max_batch {mb}): reference, {shifts}", flush=True); continue
    d = np.abs(logits - ref)
    bad = np.argwhere(d.max(axis=-1) > 0)
    if len(bad) == 0:
        print(f"run {r} (max_batch {mb}): identical, {shifts}", flush=True)
    else:
        print(f"run {r} (max_batch {mb}): {len(bad)} pixels differ, max |d| {d.max():.3e}; slices {np.unique(bad[:, 0])[:10]} "
              f"y {bad[:, 1].min()}..{bad[:, 1].max()} x {bad[:, 2].min()}..{bad[:, 2].max()}, {shifts}", flush=True)
