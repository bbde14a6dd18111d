"""Run-to-run and pass-size determinism of the parity-mode network at 512 x 512: SHA-1 of the logits of one 160-slice volume through
engines of max_batch 160 and 64 (passes of 64 + 64 + 32), several times each.  Same bits everywhere or there is a race / a dependence on
stale memory.   python scripts/unet_bits512.py [slices] [reps]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L
from dosma_amd.models import weights as W

S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hw = int(os.environ.get("BITS_HW", "512"))
w = W.random_weights(seed=0)
rng = np.random.default_rng(5120)
vol = (rng.standard_normal((S, hw, hw)) * 80 + 200).astype(np.float32)
for mb in (S, 64, S, 40):
    eng = L.Unet2dEngine(W.to_abi_order(w), hw, hw, max_batch=mb, precision="fp16x3")
    for r in range(reps):
        logits, _ = eng.forward_host(vol, whiten=True, eps=0.0)
        per_slice = [hashlib.sha1(logits[i].tobytes()).hexdigest()[:6] for i in (0, 31, 63, 64, 100, S - 1)]
        print(f"max_batch {mb:4d} run {r}: sha1 {hashlib.sha1(logits.tobytes()).hexdigest()[:16]}  slices {per_slice}", flush=True)
    eng.close()
