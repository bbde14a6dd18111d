"""Repeat one forward N times and report where runs differ from the first (a race shows up as a few pixels in a few runs).
   BITS_HW=512 python scripts/unet_race_hunt.py [slices] [reps] [max_batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L
from dosma_amd.models import weights as W

S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mbs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [40]
hw = int(os.environ.get("BITS_HW", "512"))
w = W.random_weights(seed=0)
rng = np.random.default_rng(5120)
vol = (rng.standard_normal((S, hw, hw)) * 80 + 200).astype(np.float32)
ref = None
for r in range(reps):
    mb = mbs[r % len(mbs)]
    eng = L.Unet2dEngine(W.to_abi_order(w), hw, hw, max_batch=mb, precision="fp16x3")  # fresh buffers every time
    logits, _ = eng.forward_host(vol, whiten=True, eps=0.0)
    eng.close()
    if ref is None:
        ref = logits
        print("run 0: reference", flush=True)
        continue
    d = np.abs(logits - ref)
    bad = np.argwhere(d.max(axis=-1) > 0)
    if len(bad) == 0:
        print(f"run {r} (max_batch {mb}): identical", flush=True)
    else:
        sl = np.unique(bad[:, 0]); ys = bad[:, 1]; xs = bad[:, 2]
        print(f"run {r} (max_batch {mb}): {len(bad)} pixels differ, max |d| {d.max():.3e}; slices {sl[:8]} y {ys.min()}..{ys.max()} x {xs.min()}..{xs.max()}", flush=True)
