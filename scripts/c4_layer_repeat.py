"""One convolution layer through the operator entry (qmri_conv2d_nhwc_host), many times: are the outputs bit-identical, and if not,
WHICH elements differ -- which images, rows, columns, output channels?  Default shape: up3.conv1 of the 512 x 512 network at 32
slices per pass (64 x 64 x 512 -> 256), the layer round 6's checksum bisect found first in 18 of 20 deviating forwards.

    python scripts/c4_layer_repeat.py [--B 32] [--hw 64] [--cin 512] [--cout 256] [--reps 200]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--cin", type=int, default=512)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--precision", default="fp16x3")
    a = ap.parse_args()
    from dosma_amd import _lib as L

    rng = np.random.default_rng(0)
    x = np.maximum(rng.standard_normal((a.B, a.hw, a.hw, a.cin), dtype=np.float32), 0) + 0.1
    k = (rng.standard_normal((3, 3, a.cin, a.cout), dtype=np.float32) * np.sqrt(2.0 / (9 * a.cin))).astype(np.float32)
    b = (0.1 * rng.standard_normal(a.cout)).astype(np.float32)
    ref = None
    nbad = 0
    for rep in range(a.reps):
        y = L.conv2d_nhwc_host(x, k, b, relu=True, precision=a.precision)
        if ref is None:
            ref = y.copy()
            continue
        d = y.view(np.uint32) != ref.view(np.uint32)
        if d.any():
            nbad += 1
            bi, yi, xi, ci = np.nonzero(d)
            print(f"rep {rep}: {d.sum()} elements differ | images {np.unique(bi).tolist()} rows {np.unique(yi).tolist()} cols {xi.min()}-{xi.max()} "
                  f"({len(np.unique(xi))} distinct) channels {ci.min()}-{ci.max()} ({len(np.unique(ci))} distinct; mod 32: {np.unique(ci % 32)[:8].tolist()}...) "
                  f"max|d| {np.abs(y - ref)[d].max():.3e} max|ref| {np.abs(ref).max():.2f}", flush=True)
            # per (image, row): how many columns / channels
            for bb in np.unique(bi)[:2]:
                m = bi == bb
                for yy in np.unique(yi[m])[:6]:
                    mm = m & (yi == yy)
                    print(f"    image {bb} row {yy}: cols {np.unique(xi[mm]).tolist()[:40]} channels {np.unique(ci[mm]).tolist()[:16]}... ({len(np.unique(ci[mm]))})")
    print(f"{nbad} of {a.reps - 1} repeats differ from the first")


if __name__ == "__main__":
    main()
