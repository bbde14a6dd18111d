"""The N > 1 path on CPU: two gloo processes (world_size 2), 127.0.0.1 rendezvous.

The per-volume work is stubbed with the oracle (there is no GPU here); what is under test is the
sharding (volume v -> rank v mod world), the scalar all-gather, the max-reduction of the timing, the
weight broadcast -- i.e. everything bench.py / a multi-GPU driver relies on besides the kernels."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from dosma_amd import dist as qd
from oracle import fit_oracle as fo

rank, local_rank, world = qd.init("gloo")
assert world == 2
n_vol = 5
x = np.arange(1, 5) * 10.0

def work(v):
    rng = np.random.default_rng(100 + v)
    y = rng.uniform(300, 1500, 64) * np.exp(-x[:, None] / rng.uniform(15, 80, 64))
    tc, r2, _ = fo.monoexp_fit_arrays(x, y, decimal_precision=3)
    return {"voxels": float(y.shape[1]), "mean_tc": float(tc.mean()), "rank": float(rank)}

local, summary = qd.sharded_map(n_vol, work)
assert sorted(local) == qd.partition(n_vol, world, rank)
tmax = qd.allreduce_max(1.0 + rank)
w = qd.broadcast_array(np.arange(6.0) * (1 if rank == 0 else -1))
g = qd.allgather_scalars([rank, 10 * rank])
# the configs[4] driver exactly as bench.py uses it: weights made on rank 0 and broadcast ONCE, volumes partitioned,
# per-volume work between barriers, scalars gathered
weights = {"k": np.full((3, 2), 7.0 if rank == 0 else -1.0, np.float32), "b": np.arange(4, dtype=np.float32) * (rank == 0)}
weights = qd.broadcast_weights(weights)
resident = []
def setup(vs):
    resident.extend(vs)
def vol(v):
    out = work(v)
    out["w"] = float(weights["k"][0, 0] + weights["b"][3])
    return out
batch = qd.run_batch(n_vol, vol, setup=setup)
assert resident == qd.partition(n_vol, world, rank)
qd.barrier()
print("RESULT " + json.dumps({"rank": rank, "owned": sorted(local), "tmax": tmax, "w": w.tolist(),
                              "g": g.tolist(), "summary": {k: v.tolist() for k, v in summary.items()},
                              "batch": {"per_rank": batch["per_rank"], "wall_s": batch["wall_s"], "volumes": batch["volumes"],
                                        "busy": batch["rank_busy_s"],
                                        "summary": {k: v.tolist() for k, v in batch["summary"].items()}}}))
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
        outs.append(out)
    import json

    res = [json.loads(next(l for l in o.splitlines() if l.startswith("RESULT "))[7:]) for o in outs]
    res.sort(key=lambda r: r["rank"])
    assert res[0]["owned"] == [0, 2, 4] and res[1]["owned"] == [1, 3]
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0
    assert res[0]["w"] == res[1]["w"] == list(np.arange(6.0))
    assert res[0]["g"] == res[1]["g"] == [[0.0, 0.0], [1.0, 10.0]]
    # the gathered summary is complete and identical on both ranks
    assert res[0]["summary"] == res[1]["summary"]
    assert res[0]["summary"]["rank"] == [0.0, 1.0, 0.0, 1.0, 0.0]
    assert res[0]["summary"]["voxels"] == [64.0] * 5
    assert all(15 < t < 80 for t in res[0]["summary"]["mean_tc"])
    # run_batch (BASELINE configs[4] driver): same partition, weights of rank 0 everywhere (7 + 3), one wall time
    b0, b1 = res[0]["batch"], res[1]["batch"]
    assert b0["per_rank"] == b1["per_rank"] == [3, 2] and b0["volumes"] == 5
    assert b0["summary"] == b1["summary"] and b0["summary"]["w"] == [10.0] * 5
    assert b0["summary"]["rank"] == [0.0, 1.0, 0.0, 1.0, 0.0]
    assert b0["wall_s"] == b1["wall_s"] >= max(b0["busy"]) > 0 and len(b0["busy"]) == 2


def test_partition_covers_everything_once():
    from dosma_amd.dist import partition

    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in partition(n, world, r))
            assert seen == list(range(n))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 code path exactly as the driver launches it (torch.distributed.run, one process per rank),
    with QMRI_BENCH_BACKEND=gloo so that the two ranks can share the box's single GPU: rendezvous, per-rank volumes,
    the max-over-ranks timing, the UNet weights made on rank 0 and broadcast, BASELINE configs[4] through
    dist.run_batch, the host-fed leg with every rank copying at once.  (RCCL itself needs one GPU per rank: the 8-GPU
    runs are the driver's.)"""
    import json

    env = dict(os.environ, QMRI_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--cfg5-volumes-per-gpu", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]        # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["vs_baseline"] is None
    n = 512 * 512 * 160
    assert out["config"]["voxels_per_gpu_per_step"] == n
    # value = the units ALL ranks processed / the max-over-ranks time; the N = 1-comparable run is there too
    assert abs(out["value"] - 2 * n * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert out["runs"]["A_defaults_fixed_p0"]["voxel_fits_per_s"] == out["value"]
    assert out["parity"]["nfev_equal_frac"] > 0.999 and out["parity"]["max_rel"] < 1e-4
    c5 = out["cfg5"]
    assert c5["per_rank"] == [1, 1] and c5["volumes"] == 2
    assert c5["weights_broadcast"]["identical_on_all_ranks"] and c5["weights_broadcast"]["bytes"] > 1e8
    assert c5["voxel_fits_per_s"] > 1e7 and c5["slices_per_s"] > 10
    hf = c5["host_feed"]
    assert len(hf["seconds_per_rank"]) == 2 and hf["voxel_fits_per_s"] > 1e7
    assert len(hf["int16"]["seconds_per_rank"]) == 2 and hf["int16"]["voxel_fits_per_s"] > 1e7
    for d in ("h2d", "d2h"):
        assert len(hf["copy_bandwidth"][d]["gb_per_s_per_rank"]) == 2 and hf["copy_bandwidth"][d]["gb_per_s_all_ranks"] > 1
    assert out["unet2d"]["value"] > 100 and "cpu_baseline" not in out


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher (how the driver's N = 1 command line is recorded: plain python):
    bench.py re-runs itself under torch.distributed.run, one process per rank, and this process's stdout carries rank
    0's single JSON line.  QMRI_BENCH_BACKEND=gloo lets the two ranks share the box's one GPU."""
    import json

    env = dict(os.environ, QMRI_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--cfg5-volumes-per-gpu", "1", "--no-cpu-baseline", "--no-unet"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2
    assert out["process_group"]["world_size"] == 2 and out["process_group"]["backend"] == "gloo"
    assert out["cfg5"]["per_rank"] == [1, 1]
    assert abs(out["value"] - 2 * 512 * 512 * 160 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


def test_bench_launcher_forwards_the_exit_code(tmp_path):
    """The self-launch path without a GPU: the ranks fail (no device), and the failure -- not a usage error, not a
    hang -- is what the caller sees; with WORLD_SIZE set and different from --gpus the old refusal stays."""
    env = dict(os.environ, QMRI_BENCH_BACKEND="gloo", WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 2])
def test_knee_batch_example(ranks):
    """examples/knee_batch.py (BASELINE configs[4] through the drop-in API: fit + generate_mask per volume, batch axis
    sharded over the ranks) on a small grid: as one process, and as two gloo ranks sharing the box's GPU."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    script = [os.path.join(ROOT, "examples", "knee_batch.py"), "--volumes", "3", "--shape", "64", "96", "4"]
    if ranks == 1:
        cmd = [sys.executable] + script
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + script + ["--backend", "gloo"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    head = [l for l in p.stdout.splitlines() if l.startswith("3 volumes of (64, 96, 4)")]
    assert len(head) == 1 and f"on {ranks} rank(s)" in head[0], p.stdout[-2000:]   # rank 0 reports once
    rows = {l.split()[0]: l for l in p.stdout.splitlines() if l.startswith("  ")}
    assert "voxels" in rows and "seconds" in rows and any(k.startswith("t2_mean_") for k in rows)
    # every volume was processed exactly once, whichever rank owned it: three voxel counts of 64 * 96 * 4
    assert rows["voxels"].count("24576") == 3, rows["voxels"]


RCCL_WORKER = r"""
import os, sys, json, zlib
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as td
from dosma_amd import dist as qd
from dosma_amd.models import weights as W

rank, local_rank, world = qd.init()          # launched like torch.distributed.run does it: nccl because a GPU is visible
assert world == 1 and td.is_initialized() and td.get_backend() == "nccl", (world, td.is_initialized())
assert qd._device_for_collectives().type == "cuda"
tmax = qd.allreduce_max(3.25)
g = qd.allgather_scalars([1.5, -2.0, 7.0])
wts = W.random_weights(seed=0)
before = {k: zlib.crc32(np.ascontiguousarray(v, dtype=np.float32).tobytes()) for k, v in wts.items()}
nbytes = int(sum(np.asarray(v).size for v in wts.values()) * 4)
out = qd.broadcast_weights(wts, src=0)       # ONE packed broadcast through RCCL: host -> device -> collective -> host
after = {k: zlib.crc32(np.ascontiguousarray(v, dtype=np.float32).tobytes()) for k, v in out.items()}
shapes_ok = all(out[k].shape == np.asarray(wts[k]).shape for k in wts)
resident = []
def vol(v):
    t = torch.full((1024,), float(v), device="cuda")
    return {"sum": float(t.sum().item()), "v": float(v)}
batch = qd.run_batch(3, vol, setup=resident.extend)
local, summ = qd.sharded_map(4, lambda v: {"sq": float(v * v)})
qd.barrier()
td.destroy_process_group()
print("RESULT " + json.dumps({"tmax": tmax, "g": g.tolist(), "nbytes": nbytes, "same": before == after, "shapes_ok": shapes_ok,
                              "resident": resident, "per_rank": batch["per_rank"], "sum": batch["summary"]["sum"].tolist(),
                              "wall": batch["wall_s"], "sq": summ["sq"].tolist()}))
"""


@pytest.mark.gpu
def test_rccl_world_of_one(tmp_path):
    """Every collective helper of dosma_amd.dist through the REAL "nccl" (= RCCL) backend on the box's GPU, world size 1:
    init_process_group(device_id=...), the CUDA branch of _device_for_collectives, all_reduce MAX, all_gather, the
    device round trip of the 138 MB packed weight broadcast (CRC per tensor), run_batch / sharded_map.  What an 8-GPU
    run adds to this is ranks, not code."""
    import json

    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % {"root": ROOT})
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.pop("QMRI_BENCH_BACKEND", None)
    p = subprocess.run([sys.executable, str(script)], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(next(l for l in p.stdout.splitlines() if l.startswith("RESULT "))[7:])
    assert r["tmax"] == 3.25 and r["g"] == [[1.5, -2.0, 7.0]]
    assert r["nbytes"] > 1.3e8 and r["same"] and r["shapes_ok"]
    assert r["resident"] == [0, 1, 2] and r["per_rank"] == [3] and r["sum"] == [0.0, 1024.0, 2048.0] and r["wall"] > 0
    assert r["sq"] == [0.0, 1.0, 4.0, 9.0]


@pytest.mark.gpu
def test_bench_one_rank_through_rccl():
    """bench.py launched by torch.distributed.run with ONE rank: the nccl init_process_group(device_id=...) line, the
    barriers, the MAX all-reduces of the timings and cfg5's weight broadcast + scalar all-gathers all run through RCCL
    on the leased GPU -- the first 8-GPU run is then not the first RCCL call of this code."""
    import json

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "QMRI_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--cfg5-volumes-per-gpu", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["process_group"] == {"backend": "nccl", "world_size": 1, "collectives_on": "cuda:0"}
    c5 = out["cfg5"]
    assert c5["per_rank"] == [1] and c5["weights_broadcast"]["identical_on_all_ranks"] and c5["weights_broadcast"]["bytes"] > 1.3e8
    assert out["parity"]["nfev_equal_frac"] > 0.999 and out["unet2d"]["value"] > 100
    # round 6: shader clock / board power over the timed loops, the sustained UNet leg and the reference-clock figure are in every line
    assert set(out["roofline"]["clock_samples"]) >= {"samples", "sclk_ghz_mean", "power_w_mean"} and "sclk_ghz_mean" in out["roofline"]
    u = out["unet2d"]
    assert u["sustained"]["steps"] >= 2 and u["sustained"]["slices_per_s_this_rank"] > 100 and "value_at_ref_clock" in u
    if u["roofline"]["clock_samples"]["samples"]:   # (a box that exposes the hwmon files)
        assert 0.5 < u["roofline"]["sclk_ghz_mean"] < 3.0 and 100 < u["roofline"]["power_w_mean"] < 1500
        assert abs(u["value_at_ref_clock"] - u["value"] * 1.82 / u["sustained"]["sclk_ghz_mean"]) < 1e-6 * u["value"]
    assert c5["schedule"] == "back_to_back" and c5["every_volume_once"] and c5["two_streams"]["same_results"]


@pytest.mark.gpu
def test_bench_cfg5_rehearsal_at_world_size_8():
    """BASELINE configs[4] at its REAL world size on the one GPU a test box has: `python bench.py --gpus 8`
    with NO launcher -- bench.py starts its own eight ranks under torch.distributed.run -- QMRI_BENCH_BACKEND=gloo so the ranks
    share the device, 8 volumes of 512 x 512 x 160 x 8 per rank = the 64-volume batch, fit + 512 x 512 segmentation each; the UNet
    engines take 32 slices per pass so that eight of them (and 8 x 9 resident volumes) fit one GPU's 288 GB.  What it rehearses
    for the first 8-GPU lease: the 8-process rendezvous on a free port, the weight broadcast to seven receivers (CRC-32 of every
    byte equal on all ranks), volume v -> rank v mod 8 with every index owned exactly once, eight page-locked result pools and
    eight host-fed uploads at once, the max-over-ranks timing, ONE JSON line from rank 0, exit code 0."""
    import json
    import time

    env = dict(os.environ, QMRI_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--cfg5-volumes-per-gpu", "8", "--no-cpu-baseline", "--unet-batch", "32", "--cfg5-unet-batch", "32"]
    t0 = time.time()
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    took = time.time() - t0
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["process_group"]["world_size"] == 8 and out["process_group"]["backend"] == "gloo"
    n = 512 * 512 * 160
    assert abs(out["value"] - 8 * n * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    c5 = out["cfg5"]
    assert c5["volumes"] == 64 and c5["per_rank"] == [8] * 8 and c5["schedule"] == "back_to_back"
    assert c5["volume_owner"] == [v % 8 for v in range(64)] and c5["every_volume_once"]
    wb = c5["weights_broadcast"]
    assert wb["identical_on_all_ranks"] and len(wb["crc32_per_rank"]) == 8 and len(set(wb["crc32_per_rank"])) == 1
    assert wb["bytes"] > 1.3e8
    assert c5["two_streams"]["same_results"]
    assert len(c5["rank_busy_s"]) == 8 and c5["wall_s"] >= max(c5["rank_busy_s"]) > 0
    hf = c5["host_feed"]
    assert len(hf["seconds_per_rank"]) == 8 and len(hf["int16"]["seconds_per_rank"]) == 8
    assert len(hf["copy_bandwidth"]["h2d"]["gb_per_s_per_rank"]) == 8
    assert out["parity"]["nfev_equal_frac"] > 0.999 and out["parity"]["max_rel"] < 1e-4
    assert out["unet2d"]["value"] > 100 and "cpu_baseline" not in out
    assert took < 240, f"{took:.0f} s"
