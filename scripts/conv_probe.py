"""One convolution layer through the operator-level entry, for rocprofv3 (kernel trace / PMC) on a single kernel.

    python scripts/conv_probe.py --cin 128 --cout 128 --hw 96 96 --batch 32 [--precision fp16x3] [--transposed]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dosma_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=128)
ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--hw", type=int, nargs=2, default=[96, 96])
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--precision", default="fp16x3")
ap.add_argument("--transposed", action="store_true")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
rng = np.random.default_rng(0)
x = rng.standard_normal((a.batch, a.hw[0], a.hw[1], a.cin)).astype(np.float32)
k = (rng.standard_normal((3, 3, a.cout, a.cin) if a.transposed else (3, 3, a.cin, a.cout)) / np.sqrt(9 * a.cin)).astype(np.float32)
b = rng.standard_normal(a.cout).astype(np.float32)
for _ in range(a.reps):
    y = L.conv2d_nhwc_host(x, k, b, relu=True, transposed=a.transposed, precision=a.precision)
print("ok", y.shape, float(np.abs(y).mean()))
