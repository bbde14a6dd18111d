"""Time the U-Net forward (device-resident input) at the bench size."""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L
from dosma_amd.models import weights as W
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, default=384)
ap.add_argument("--slices", type=int, default=64)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
w = W.random_weights(seed=0)
# QMRI_PROF_BN_SHIFT0=1: BatchNorm without shift (moving_mean = beta = 0): the normalised tensors are scale * relu(.) -- half
# zeros -- instead of dense.  Same kernels, same work; what changes is the data the matrix pipes toggle on, i.e. the power the
# forward draws against the board's cap.  The A/B prices a network that DEFERS the shift into its consumers.
if os.environ.get("QMRI_PROF_BN_SHIFT0", "0") == "1":
    for k in w:
        if k.endswith("_bn_mean") or k.endswith("_bn_beta"):
            w[k] = np.zeros_like(w[k])
eng = L.Unet2dEngine(W.to_abi_order(w), args.hw, args.hw, max_batch=args.batch, precision=args.precision.split(",")[0])
dev = torch.device("cuda", 0)
x = torch.randn((args.slices, args.hw, args.hw), device=dev)
logits = torch.empty((args.slices, args.hw, args.hw, 4), device=dev)
mask = torch.empty((args.slices, args.hw, args.hw, 4), device=dev, dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
for prec in args.precision.split(","):
    eng.set_precision(prec)
    ts = []
    for i in range(args.reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.forward_device(x.data_ptr(), args.slices, logits.data_ptr(), mask.data_ptr(), whiten=True, stream=st)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    best = min(ts)
    gflop = 70.79 * (args.hw / 384) ** 2
    print(f"{prec}: {args.slices} slices {args.hw}^2 batch {args.batch}: best {best*1e3:.1f} ms -> {args.slices/best:.1f} slices/s, {gflop*args.slices/best/1e3:.1f} TFLOP/s  all {[round(t*1e3,1) for t in ts]}")
