"""Which LAYER first differs when a forward of the same volume comes out different?  (QMRI_UNET_CHECKSUMS=1: the engine queues an
exact checksum of every layer's output buffer in every pass; qmri_unet2d_trace returns them.)

    QMRI_UNET_CHECKSUMS=1 python scripts/unet_layer_bisect.py [--batch 32] [--slices 160] [--hw 512] [--reps 300]
"""
import argparse
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QMRI_UNET_CHECKSUMS", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--slices", type=int, default=160)
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    import torch

    import bench
    from dosma_amd import _lib as L
    from dosma_amd.models import weights as W

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    H, S = args.hw, args.slices
    eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), H, H, max_batch=args.batch, precision="fp16x3", device=0)
    y = bench.make_volume(torch, dev, 20260928)
    x = y[0][: S * H * H]
    st = torch.cuda.current_stream(dev)
    logits = torch.empty((S, H, H, 4), device=dev)
    mask = torch.empty((S, H, H, 4), device=dev, dtype=torch.uint8)

    def sums():
        buf = ctypes.create_string_buffer(1 << 20)
        eng._lib.qmri_unet2d_trace(eng._handle, buf, len(buf))
        out = collections.OrderedDict()
        for t in buf.value.decode().split(";"):
            if t.startswith("#"):
                k, v = t.split("=")
                out[k] = v
        return out

    ref = None
    first_bad = collections.Counter()
    for rep in range(args.reps):
        eng.forward_device(x.data_ptr(), S, logits.data_ptr(), mask.data_ptr(), whiten=True, stream=st.cuda_stream)
        torch.cuda.synchronize()
        cs = sums()
        if rep == 0:
            continue  # (the first forward sees never-written halves of the concat buffers: the second one is the reference)
        if ref is None:
            ref = cs
            print(len(cs), "checksums per forward; layers of pass 0:", [k for k in cs if k.startswith("#0.")])
            continue
        bad = [k for k in cs if cs[k] != ref.get(k)]
        if bad:
            first_bad[bad[0].split(".", 1)[1]] += 1
            print(f"rep {rep}: {len(bad)} checksums differ; in order: {bad[:14]}", flush=True)
    print("first differing layer, counted over the bad forwards:", dict(first_bad))
    eng.close()


if __name__ == "__main__":
    main()
