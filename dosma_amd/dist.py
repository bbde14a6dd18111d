"""Multi-GPU driver: batches of volumes sharded on the batch axis, one process per GPU.

The reference has no multi-GPU or distributed path at all (grep for nccl / mpi /
torch.distributed -> 0 hits; its only parallelism is ``multiprocessing.Pool`` over voxels,
dosma/core/fitting.py:860-868).  The path shards embarrassingly: voxels, slices and
volumes are independent, so volume ``v`` of a batch goes to rank ``v mod world`` and NO collective sits
on the data path.  RCCL (``torch.distributed`` backend "nccl" on ROCm) / gloo is used only for
(a) the barrier + max-reduction that brackets timing, (b) an all-gather of a few per-rank scalars
(voxel counts, elapsed time, per-volume summary statistics), (c) optionally a one-time broadcast of
model weights.  Launch with ``python -m torch.distributed.run --nproc-per-node N ...``.
"""
import os
from typing import Callable, Dict, List, Sequence

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def launched_by_torchrun() -> bool:
    """True under ``python -m torch.distributed.run`` (also with ``--nproc-per-node 1``): it exports the rendezvous."""
    return all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))


def init(backend: str = None, force: bool = False):
    """Initialise the process group.  Returns (rank, local_rank, world).

    A plain single process (``python script.py``) stays without a group: every helper below then degenerates to the
    identity.  Under ``torch.distributed.run`` -- with ANY number of ranks, one included -- or with ``force=True`` the
    group is created, so that a world of one runs the same RCCL calls an 8-GPU node will (tests/test_dist.py).

    backend: "nccl" (= RCCL over xGMI on ROCm) when a GPU is visible, else "gloo".
    """
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if (world > 1 or force or launched_by_torchrun()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    if torch.cuda.is_available():  # one process per GPU: this rank's host-level calls go to its own device
        from dosma_amd import _lib

        _lib.set_default_device(local_rank % max(torch.cuda.device_count(), 1))
    return rank, local_rank, world


def partition(n_items: int, world: int, rank: int) -> List[int]:
    """Indices of the batch items owned by ``rank``: item v -> rank v mod world."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    return list(range(rank, n_items, world))


def _device_for_collectives():
    import torch
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier():
    import torch.distributed as dist

    if dist.is_initialized():
        dist.barrier()


def allreduce_max(value: float) -> float:
    """Max over ranks (used for the elapsed time of a timed region)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_device_for_collectives())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_scalars(values: Sequence[float]) -> np.ndarray:
    """All-gather a short, fixed-length vector of float64 per rank -> array (world, len(values))."""
    import torch
    import torch.distributed as dist

    v = np.asarray(values, dtype=np.float64).reshape(-1)
    if not dist.is_initialized():
        return v[None, :].copy()
    dev = _device_for_collectives()
    mine = torch.tensor(v, dtype=torch.float64, device=dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return np.stack([o.cpu().numpy() for o in out], axis=0)


def _gather_item_scalars(n_items: int, mine: List[int], local: Dict[int, Dict[str, float]], world: int) -> Dict[str, np.ndarray]:
    """All-gather the per-item scalar dictionaries: {key: ndarray[n_items]}, identical on every rank.  Every rank
    contributes a vector of the same length (items per rank padded to the maximum, NaN index = no item)."""
    keys = sorted(next(iter(local.values())).keys()) if local else []
    per_rank = (n_items + world - 1) // world
    nk = int(allreduce_max(float(len(keys))))
    if not keys:
        keys = [f"_{i}" for i in range(nk)]
    buf = np.full((per_rank, 1 + nk), np.nan)
    for slot, v in enumerate(mine):
        buf[slot, 0] = v
        buf[slot, 1:] = [local[v][k] for k in keys]
    gathered = allgather_scalars(buf.reshape(-1)).reshape(world, per_rank, 1 + nk)
    summary = {k: np.full(n_items, np.nan) for k in keys}
    for r in range(world):
        for slot in range(per_rank):
            idx = gathered[r, slot, 0]
            if not np.isnan(idx):
                for j, k in enumerate(keys):
                    summary[k][int(idx)] = gathered[r, slot, 1 + j]
    return summary


def sharded_map(n_items: int, fn: Callable[[int], Dict[str, float]]):
    """Run ``fn(v)`` for every batch item this rank owns; gather every item's scalar summary everywhere.

    ``fn`` does the per-volume work on this rank's GPU (e.g. ``MonoExponentialFit.fit`` + ``generate_mask``)
    and returns a dict of scalars (same keys for every item), e.g. voxel counts and mean tc.
    Returns (local: {v: fn(v)}, summary: {key: ndarray[n_items]}) -- the summary is identical on all ranks.
    """
    rank, _, world = env_world()
    mine = partition(n_items, world, rank)
    local = {v: fn(v) for v in mine}
    return local, _gather_item_scalars(n_items, mine, local, world)


def broadcast_array(arr: np.ndarray, src: int = 0) -> np.ndarray:
    """One-time broadcast of a (weights) array from ``src`` to every rank."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return arr
    dev = _device_for_collectives()
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def broadcast_weights(weights: Dict[str, np.ndarray], src: int = 0) -> Dict[str, np.ndarray]:
    """One-time broadcast of a model's weight dictionary from ``src``: the arrays are packed into ONE contiguous
    float32 buffer so that the exchange is a single collective (RCCL over xGMI with the "nccl" backend; 138 MB for the
    6-level U-Net), instead of one small broadcast per tensor.  Names / shapes are rank-local knowledge of the
    architecture: every rank passes a dictionary with the same keys and shapes (its values are ignored off ``src``)."""
    import torch.distributed as dist

    if not dist.is_initialized():
        return weights
    keys = sorted(weights)
    sizes = [int(np.prod(weights[k].shape)) for k in keys]
    flat = np.empty(sum(sizes), dtype=np.float32)
    if dist.get_rank() == src:
        off = 0
        for k, n in zip(keys, sizes):
            flat[off:off + n] = np.asarray(weights[k], dtype=np.float32).reshape(-1)
            off += n
    flat = broadcast_array(flat, src=src)
    out, off = {}, 0
    for k, n in zip(keys, sizes):
        out[k] = flat[off:off + n].reshape(weights[k].shape).copy()
        off += n
    return out


def run_batch(n_volumes: int, per_volume: Callable[[int], Dict[str, float]], *, setup: Callable[[List[int]], None] = None,
              pipelined: Callable[[List[int]], Dict[int, Dict[str, float]]] = None):
    """BASELINE.json configs[4] as a driver: a batch of ``n_volumes`` independent volumes over the ranks of one node.

    Volume v -> rank v mod world (:func:`partition`); ``setup(my_volumes)`` (optional) runs before the clock starts
    (e.g. making the rank's inputs resident in HBM); then, between two barriers, every rank runs
    ``per_volume(v)`` for its volumes -- the hot path on its own GPU, no collective inside -- and the wall time is the
    MAX over ranks.  The per-volume scalars are all-gathered once at the end (control plane).  ``pipelined(my_volumes)``
    (optional) replaces the per-volume loop by the caller's own schedule over the rank's volumes (e.g. the fit of volume
    v + 1 on a second stream under the segmentation of volume v) and returns the same per-volume scalars.

    Returns dict(wall_s, volumes, volumes_per_s, per_rank=[n volumes], summary={key: ndarray[n_volumes]}).
    """
    import time

    rank, _, world = env_world()
    mine = partition(n_volumes, world, rank)
    if setup is not None:
        setup(mine)
    barrier()
    t0 = time.perf_counter()
    local = pipelined(mine) if pipelined is not None else {v: per_volume(v) for v in mine}
    mine_s = time.perf_counter() - t0
    barrier()
    wall = allreduce_max(time.perf_counter() - t0)
    summary = _gather_item_scalars(n_volumes, mine, local, world)  # control plane, once, after the clock
    busy = allgather_scalars([mine_s])[:, 0]
    return {"wall_s": wall, "volumes": n_volumes, "volumes_per_s": n_volumes / wall if wall > 0 else float("inf"),
            "per_rank": [len(partition(n_volumes, world, r)) for r in range(world)],
            "rank_busy_s": busy.tolist(), "summary": summary}
