"""Round 6: the FIRST forward of a fresh engine whose volume takes several passes (slices > max_batch) differed from every later one
(scripts/unet_concurrency_check.py, batch 32: one slice off by up to 0.13).  Localise: which (slices, max_batch, size) combinations,
which slices of which pass, and whether poisoning freshly allocated device memory with NaN patterns makes the reads visible.

    python scripts/unet_firstcall_check.py [--poison]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--cases", default="512:160:32,512:64:32,512:32:32,512:160:64,512:96:64,384:160:32,384:160:160,512:160:160")
    args = ap.parse_args()
    import torch

    from dosma_amd import _lib as L
    from dosma_amd.models import weights as W

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wts = W.to_abi_order(W.random_weights(seed=0))
    gen = torch.Generator(device=dev).manual_seed(3)
    for case in args.cases.split(","):
        hw, S, B = (int(v) for v in case.split(":"))
        if args.poison:  # fill ~60 GB of free device memory with a NaN bit pattern (fp32 and fp16 NaN alike), then release it
            junk = [torch.full((1 << 30,), -1, dtype=torch.int32, device=dev) for _ in range(15)]
            torch.cuda.synchronize()
            del junk
            torch.cuda.empty_cache()
        x = torch.randn((S, hw, hw), device=dev, generator=gen) * 150 + 300
        eng = L.Unet2dEngine(wts, hw, hw, max_batch=B, precision="fp16x3", device=0)
        st = torch.cuda.current_stream(dev)
        outs = []
        for rep in range(3):
            lg = torch.empty((S, hw, hw, 4), device=dev)
            mk = torch.empty((S, hw, hw, 4), device=dev, dtype=torch.uint8)
            eng.forward_device(x.data_ptr(), S, lg.data_ptr(), mk.data_ptr(), whiten=True, stream=st.cuda_stream)
            torch.cuda.synchronize()
            outs.append(lg)
        line = f"{hw}x{hw} S={S} max_batch={B}:"
        for a, b, tag in ((0, 1, "1st vs 2nd"), (1, 2, "2nd vs 3rd")):
            d = outs[a].view(torch.int32) != outs[b].view(torch.int32)
            per_slice = d.flatten(1).sum(1).cpu().numpy()
            bad = np.flatnonzero(per_slice)
            line += f" [{tag}: {int(d.sum().item())} differ, slices {bad[:10].tolist()}, NaN {int(torch.isnan(outs[a]).sum().item())}"
            if len(bad):
                s = int(bad[0])
                dd = d[s].any(-1)
                ys, xs = dd.nonzero(as_tuple=True)
                line += f", slice {s}: rows {int(ys.min())}-{int(ys.max())} cols {int(xs.min())}-{int(xs.max())} n={int(dd.sum())}, max|d| {float((outs[a][s] - outs[b][s]).abs().max()):.2e}"
            line += "]"
        print(line, flush=True)
        eng.close()
        del outs, x


if __name__ == "__main__":
    main()
