"""BASELINE.json configs[4]: a batch of knee volumes -- mono-exponential T2 fit + 2D-UNet segmentation per volume --
sharded on the batch axis over the GPUs of one node (one process per GPU, no collective on the data path).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        examples/knee_batch.py --volumes 64

Volume v goes to rank v mod world (dosma_amd.dist.partition).  Per volume the rank runs, through the drop-in API on its
own GPU, ``MonoExponentialFit.fit`` on the 8 echoes and ``IWOAIOAIUnet2DNormalized.generate_mask`` on the first echo,
and reports per-tissue mean T2 like ``QuantitativeValue.to_metrics``; one all-gather of those scalars at the end is the
only communication (RCCL over xGMI with the "nccl" backend; ``--backend gloo`` lets ranks share a GPU for testing).
Data are synthetic (no network for datasets); weights are seeded random in the reference's architecture.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import dosma_amd as dm  # noqa: E402
from dosma_amd import dist  # noqa: E402
from dosma_amd.models import IWOAIOAIUnet2DNormalized  # noqa: E402
from dosma_amd.models import weights as W  # noqa: E402


# (SI, AP, LR) voxel grid: 0.4 mm in plane, 1.5 mm slices -- the orientation the segmentation models work in
SAGITTAL_AFFINE = np.array([[0, 0, 1.5, 0.0], [0, -0.4, 0, 0.0], [-0.4, 0, 0, 0.0], [0, 0, 0, 1.0]])


def synthetic_knee(v, shape, te):
    rng = np.random.default_rng(1000 + v)
    t2 = rng.uniform(20, 70, shape).astype(np.float32)
    s0 = rng.uniform(300, 1500, shape).astype(np.float32)
    vols = []
    for t in te:
        e = s0 * np.exp(-np.float32(t) / t2) + rng.standard_normal(shape, dtype=np.float32) * 12
        vols.append(dm.MedicalVolume(e.astype(np.float32), SAGITTAL_AFFINE))
    return vols


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volumes", type=int, default=8)
    ap.add_argument("--shape", type=int, nargs=3, default=[384, 384, 160])
    ap.add_argument("--backend", default=None, help="nccl (default on GPUs) | gloo")
    ap.add_argument("--precision", default="fp16x3")
    args = ap.parse_args()
    rank, local_rank, world = dist.init(args.backend)
    shape = tuple(args.shape)
    te = np.arange(1, 9) * 10.0
    IWOAIOAIUnet2DNormalized.precision = args.precision
    model = IWOAIOAIUnet2DNormalized((shape[0], shape[1], 1), W.random_weights(seed=3), force_weights=True)
    fitter = dm.MonoExponentialFit(tc0="polyfit", decimal_precision=3)

    def per_volume(v):
        echoes = synthetic_knee(v, shape, te)
        t0 = time.perf_counter()
        t2map, r2 = fitter.fit(te, echoes)
        masks = model.generate_mask(echoes[0])
        dt = time.perf_counter() - t0
        out = {"seconds": dt, "voxels": float(np.prod(shape))}
        for name, m in masks.items():
            sel = (m.volume > 0) & (t2map.volume > 0)
            out[f"t2_mean_{name}"] = float(t2map.volume[sel].mean()) if sel.any() else float("nan")
            out[f"voxels_{name}"] = float(sel.sum())
        return out

    dist.barrier()
    t0 = time.perf_counter()
    local, summary = dist.sharded_map(args.volumes, per_volume)
    dist.barrier()
    wall = dist.allreduce_max(time.perf_counter() - t0)
    if rank == 0:
        print(f"{args.volumes} volumes of {shape} x 8 echoes on {world} rank(s): {wall:.2f} s wall "
              f"({args.volumes / wall:.2f} volumes/s incl. synthetic data generation)")
        for k in sorted(summary):
            print(f"  {k:18s} {np.array2string(summary[k][:8], precision=2)}")


if __name__ == "__main__":
    main()
