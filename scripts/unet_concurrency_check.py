"""Is the parity-mode forward bit-identical when another kernel (the fit, on a second stream) runs beside it?

Round 6: bench.py's cfg5 two-stream schedule with the engine at 32 slices per pass gave 27 of 4.2e7 mask voxels different from the
back-to-back schedule on the first volume (gpurun_out/r06b/b32.json); at 160 slices per pass it did not.  This script reproduces
the situation on one volume and localises what differs: which slices / pixels / classes, NaN or last-bit.

    python scripts/unet_concurrency_check.py [--batch 32] [--slices 160] [--hw 512] [--reps 6] [--side fit|matmul|none]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--slices", type=int, default=160)
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--side", default="fit")
    ap.add_argument("--precision", default="fp16x3")
    args = ap.parse_args()
    import torch

    import bench
    from dosma_amd import _lib as L
    from dosma_amd.models import weights as W

    lib = L.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    H = args.hw
    S = args.slices
    eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), H, H, max_batch=args.batch, precision=args.precision, device=0)
    y = bench.make_volume(torch, dev, 20260928)  # (8, 512*512*160) -- the fit's input AND (echo 0, first S*H*H values) the network's
    x_ptr = y[0].data_ptr()
    n = y.shape[1]
    logits = torch.empty((S, H, H, 4), device=dev)
    mask = torch.empty((S, H, H, 4), device=dev, dtype=torch.uint8)
    main_s = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    popt = torch.empty((n, 2), dtype=torch.float32, device=dev)
    r2 = torch.empty(n, dtype=torch.float32, device=dev)
    a_mat = torch.randn((8192, 8192), device=dev, dtype=torch.float16)

    def forward():
        eng.forward_device(x_ptr, S, logits.data_ptr(), mask.data_ptr(), whiten=True, stream=main_s.cuda_stream)

    def side_work():
        if args.side == "fit":
            for _ in range(2):
                a = bench.make_args(L, y, popt, r2, side.cuda_stream, "A")
                a.device = 0
                L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
        elif args.side == "matmul":
            with torch.cuda.stream(side):
                for _ in range(30):
                    a_mat @ a_mat

    forward()
    torch.cuda.synchronize()
    ref_l, ref_m = logits.clone(), mask.clone()
    print("trace:", [t for t in eng.trace() if "act_shift" in t or "down0" in t or "up0" in t])
    forward()
    torch.cuda.synchronize()
    print("alone again: logits equal", bool(torch.equal(logits.view(torch.int32), ref_l.view(torch.int32))), "mask equal", bool(torch.equal(mask, ref_m)))
    for rep in range(args.reps):
        logits.fill_(123.0)
        mask.fill_(9)
        torch.cuda.synchronize()
        side_work()
        forward()
        torch.cuda.synchronize()
        dl = logits.view(torch.int32) != ref_l.view(torch.int32)
        dm = mask != ref_m
        nd, nm = int(dl.sum().item()), int(dm.sum().item())
        line = f"rep {rep} side={args.side}: logits differing {nd}, mask differing {nm}, NaN {int(torch.isnan(logits).sum().item())}"
        if nd:
            idx = dl.nonzero()[:2000].cpu().numpy()
            sl = np.unique(idx[:, 0])
            maxabs = float((logits - ref_l).abs()[dl].max().item())
            line += f" | slices {sl[:20].tolist()} rows {np.unique(idx[:, 1])[:12].tolist()} cols {np.unique(idx[:, 2])[:12].tolist()} classes {np.unique(idx[:, 3]).tolist()} max |d| {maxabs:.3e}"
            line += " | " + ",".join(t for t in eng.trace() if "act_shift" in t)
        print(line, flush=True)
    eng.close()


if __name__ == "__main__":
    main()
