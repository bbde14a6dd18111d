"""N forwards of the same volume through one engine: how many distinct logit tensors come out?  (Round 6: at 32 slices per pass the
bench volume gave two classes within one process -- see DESIGN section 6.6.)  Run under the kernel-family switches to find the layer.

    QMRI_C4=0 python scripts/unet_repeat_check.py [--batch 32] [--slices 160] [--hw 512] [--reps 8] [--input bench|randn]
"""
import argparse
import hashlib
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
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--input", default="bench")
    ap.add_argument("--churn", type=int, default=1, help="allocate / free device memory between forwards")
    ap.add_argument("--precision", default="fp16x3")
    args = ap.parse_args()
    import torch

    import bench
    from dosma_amd import _lib as L
    from dosma_amd.models import weights as W

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    H, S = args.hw, args.slices
    eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), H, H, max_batch=args.batch, precision=args.precision, device=0)
    if args.input == "bench" and S * H * H <= 512 * 512 * 160:
        y = bench.make_volume(torch, dev, 20260928)
        x = y[0][: S * H * H]
    else:
        x = torch.randn(S * H * H, device=dev) * 150 + 300
    st = torch.cuda.current_stream(dev)
    logits = torch.empty((S, H, H, 4), device=dev)
    mask = torch.empty((S, H, H, 4), device=dev, dtype=torch.uint8)
    keep = []
    classes = {}
    first = None
    for rep in range(args.reps):
        eng.forward_device(x.data_ptr(), S, logits.data_ptr(), mask.data_ptr(), whiten=True, stream=st.cuda_stream)
        torch.cuda.synchronize()
        per_slice = logits.view(torch.int32).flatten(1).to(torch.int64).sum(1).cpu().numpy()
        key = hashlib.sha1(per_slice.tobytes()).hexdigest()[:10]
        if first is None:
            first = per_slice
            ref = logits.clone()
        bad = np.flatnonzero(per_slice != first)
        classes.setdefault(key, []).append(rep)
        extra = ""
        if len(bad):  # where in the slices: rows / columns / how many pixels / how far
            for sl in bad[:3]:
                d = (logits[sl].view(torch.int32) != ref[sl].view(torch.int32)).any(-1)
                ys, xs = d.nonzero(as_tuple=True)
                extra += f" | s{sl}: rows {int(ys.min())}-{int(ys.max())} cols {int(xs.min())}-{int(xs.max())} n={int(d.sum())} max|d|={float((logits[sl] - ref[sl]).abs().max()):.1e}"
        print(f"rep {rep}: class {key} slices differing from rep 0: {bad[:12].tolist()}{' ...' if len(bad) > 12 else ''} ({len(bad)}){extra}", flush=True)
        if args.churn:
            keep.append(logits.clone() if rep % 2 == 0 else torch.empty(1 << (20 + rep % 8), device=dev))
            if rep % 3 == 2:
                keep.clear()
    print("env", {k: v for k, v in os.environ.items() if k.startswith("QMRI_")}, "->", len(classes), "distinct results:", classes)
    eng.close()


if __name__ == "__main__":
    main()
