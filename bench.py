#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native qMRI hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver through torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): voxel-fits/sec, 8-echo mono-exponential T2 fit of a 512x512x160 volume
(config[1]: "T2 monoexponential fit, 512x512x160 x 8 echoes, fp32, 1 MI355X").

A "step" is one pass of the hot path over one synthetic volume per GPU: ONE launch of the fused
kernel (LM fit + MonoExponentialFit post-processing) on (8, 41 943 040) fp32 samples already
resident in HBM, writing fp32 (a, tc) + r2 -- i.e. `MonoExponentialFit().fit(x, y)` with the
reference's defaults (tc0 = 30 -> p0 = (1, -1/30), bounds (0, 100), r2 >= 0.9, 1 decimal).
The scan-class recipe (tc0 = "polyfit", 3 decimals) is timed too and reported under "runs".
Multi-GPU: volumes are independent, so each rank fits its own volume (weak scaling, no data-path
collective); the only communication is the barrier / max-reduction of the timing.

Rank 0 prints ONE JSON line (see the task contract) including
  "roofline":     algorithmic HBM bytes (44 B/voxel = 8 x 4 B in + 12 B out) / kernel
                  time measured with HIP events on the launch stream, vs the 8 TB/s HBM3E peak;
  "cpu_baseline": the reference's own call pattern (one scipy.optimize.curve_fit per voxel under
                  multiprocessing.Pool, dosma/core/fitting.py:855-868, 1026-1073) timed on this
                  host's cores on a bounded sample of the same volume (N = 1 runs only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPE = (512, 512, 160)
E = 8
TE = np.arange(1, E + 1) * 10.0  # ms
BYTES_PER_VOXEL = 4 * E + 4 * 3  # fp32 echoes in, fp32 (a, tc, r2) out -- SURVEY.md section 8(d)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_volume(torch, device, seed):
    """70 % tissue S0~U(300,1500), T2~U(15,80) ms, SNR 50; 30 % exact zeros."""
    n = SHAPE[0] * SHAPE[1] * SHAPE[2]
    gen = torch.Generator(device=device).manual_seed(seed)
    s0 = torch.rand(n, device=device, generator=gen) * 1200 + 300
    t2 = torch.rand(n, device=device, generator=gen) * 65 + 15
    te = torch.tensor(TE, device=device, dtype=torch.float32)
    y = s0[None, :] * torch.exp(-te[:, None] / t2[None, :])
    y += (900.0 / 50.0) * torch.randn(y.shape, device=device, generator=gen)
    bg = torch.rand(n, device=device, generator=gen) < 0.3
    y[:, bg] = 0
    return y.contiguous()


def make_args(L, y, popt, r2, stream, recipe):
    a = L.default_args()
    n = y.shape[1]
    a.y, a.y_dtype, a.E, a.N, a.ld = y.data_ptr(), L.QMRI_F32, E, n, n
    a.x = TE.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    a.popt, a.r2, a.out_dtype = popt.data_ptr(), r2.data_ptr(), L.QMRI_F32
    a.stream = stream
    bounds = ((-np.inf, np.inf), (0.0, 100.0))
    if recipe == "A":  # MonoExponentialFit() defaults
        a.init, a.a0, a.b0 = L.INIT_SCALAR, 1.0, -1 / 30.0
        L.set_post(a, inv_abs_b=True, bounds=bounds, r2_threshold=0.9, nan_to_num=0.0, decimals=1)
    else:              # scan classes: tc0="polyfit", decimal_precision=3
        a.init = L.INIT_LOGLIN
        L.set_post(a, inv_abs_b=True, bounds=bounds, r2_threshold=0.9, nan_to_num=0.0, decimals=3)
    return a


def bench_t1rho_roi(L, lib, torch, device, local_rank, args):
    """BASELINE.json configs[2] (a parity-test configuration, reported for information): CubeQuant T1rho, 4 spin-lock
    times, 384 x 384 x 120 int16 volumes, cartilage-mask ROI (~2 % of the voxels), tc0="polyfit", bounds (0, 500),
    3 decimals.  Device-resident, kernel time from HIP events."""
    shape = (384, 384, 120)
    n = shape[0] * shape[1] * shape[2]
    tsl = np.array([1.0, 10.0, 30.0, 60.0])
    gen = torch.Generator(device=device).manual_seed(384)
    s0 = torch.rand(n, device=device, generator=gen) * 2200 + 800
    t1r = torch.rand(n, device=device, generator=gen) * 45 + 25
    t = torch.tensor(tsl, device=device, dtype=torch.float32)
    y = s0[None, :] * torch.exp(-t[:, None] / t1r[None, :]) + 20.0 * torch.randn((4, n), device=device, generator=gen)
    y = y.round().clamp(-32768, 32767).to(torch.int16).contiguous()
    mask = torch.zeros(shape, dtype=torch.uint8, device=device)
    mask[150:230, 120:260, 40:70] = (torch.rand((80, 140, 30), device=device, generator=gen) < 0.85).to(torch.uint8)
    mask = mask.reshape(-1).contiguous()
    roi = int(mask.sum().item())
    popt = torch.empty((n, 2), dtype=torch.float32, device=device)
    r2 = torch.empty(n, dtype=torch.float32, device=device)
    tc = torch.empty(n, dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream(device)
    a = L.default_args()
    a.y, a.y_dtype, a.E, a.N, a.ld = y.data_ptr(), L.QMRI_I16, 4, n, n
    a.x = tsl.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    a.mask = mask.data_ptr()
    a.popt, a.r2, a.tc, a.out_dtype = popt.data_ptr(), r2.data_ptr(), tc.data_ptr(), L.QMRI_F32
    a.stream = stream.cuda_stream
    a.device = local_rank
    a.init = L.INIT_LOGLIN
    L.set_post(a, inv_abs_b=True, bounds=((-np.inf, np.inf), (0.0, 500.0)), r2_threshold=0.9, nan_to_num=0.0, decimals=3)
    for _ in range(max(args.warmup, 1)):
        L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    ev0.record(stream)
    for _ in range(args.steps):
        L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
    ev1.record(stream)
    torch.cuda.synchronize(device)
    ms = ev0.elapsed_time(ev1) / args.steps
    return {"config": "T1rho (CubeQuant recipe), 4 spin-lock times, 384x384x120 int16, ROI mask only (BASELINE configs[2])",
            "volume_voxels": n, "roi_voxels": roi, "kernel_ms": ms,
            "roi_voxel_fits_per_s": roi / (ms * 1e-3), "volume_voxels_per_s": n / (ms * 1e-3),
            "hbm_gb_per_s": (n * (1 + 3 * 4) + roi * 4 * 2) / (ms * 1e-3) / 1e9,
            "kernel": lib.qmri_monoexp_kernel_name(ctypes.byref(a)).decode()}


def cpu_baseline(y_dev, cores_cap=None):
    """The reference's per-voxel scipy loop under multiprocessing.Pool(all cores) (fitting.py:855-868)
    on a bounded sample of the bench volume: ~6000 voxels per core (a few seconds per core)."""
    import multiprocessing as mp

    import scipy

    from oracle import fit_oracle as fo  # the checker, used here only as the timed CPU baseline

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    if cores_cap:
        cores = min(cores, cores_cap)
    n = int(min(y_dev.shape[1], 6000 * cores, 2_000_000))
    ys = y_dev[:, :n].cpu().numpy()
    p0 = (1.0, -1 / 30.0)
    jobs = [(fo.monoexponential, TE, ys[:, i], p0, fo.FTOL, fo.MAXFEV, fo.R2_EPS, 2, False)
            for i in range(n)]
    with mp.Pool(cores) as pool:
        pool.map(fo._one_voxel_scipy, jobs[: cores * 4], chunksize=4)  # start the workers
        t = time.perf_counter()
        pool.map(fo._one_voxel_scipy, jobs, chunksize=1000)
        dt = time.perf_counter() - t
    # single-thread C restatement of MINPACK on a slice, for scale
    m = min(n, 200_000)
    t = time.perf_counter()
    fo.curve_fit_c(TE, ys[:, :m], p0, threads=1)
    dt_c = time.perf_counter() - t
    # the same C restatement on every core (pthreads inside the library): the best this host can do with the algorithm
    fo.curve_fit_c(TE, ys[:, : min(n, 4 * cores)], p0, threads=cores)  # start the threads
    t = time.perf_counter()
    fo.curve_fit_c(TE, ys, p0, threads=cores)
    dt_call = time.perf_counter() - t
    return {
        "value": n / dt, "unit": "voxel-fits/s", "cores": cores, "kind": "port",
        "sample": (f"first {n} voxels of the bench volume (70% tissue / 30% zero background): one "
                   f"scipy.optimize.curve_fit per voxel under multiprocessing.Pool({cores}), "
                   f"chunksize 1000 (the reference's call pattern), scipy {scipy.__version__}, "
                   f"{dt:.1f} s wall, pool start-up excluded"),
        "per_core": n / dt / cores,
        "c_restatement_1thread": m / dt_c,
        "c_restatement_all_cores": n / dt_call,
    }


def bench_dess(L, lib, torch, device, local_rank, world, args, barrier):
    """qDESS analytic T2 map: one fused streaming pass over two fp32 echo volumes of
    384x384x160 -> fp64 map (the reference's dtype).  HBM-bound: 2 x 4 B in + 8 B out per voxel."""
    n = 384 * 384 * 160
    gen = torch.Generator(device=device).manual_seed(11 + local_rank)
    e1 = torch.rand(n, device=device, generator=gen) * 780 + 20
    e2 = e1 * (torch.rand(n, device=device, generator=gen) * 0.88 + 0.02)
    t2 = torch.empty(n, device=device, dtype=torch.float64)
    a = L.QmriDessArgs()
    a.echo1, a.echo2, a.dtype, a.out_dtype, a.N = e1.data_ptr(), e2.data_ptr(), L.QMRI_F32, L.QMRI_F64, n
    a.c0, a.k, a.c1 = -27.864, 0.0434, 3.9e-3
    a.use_bounds, a.lo, a.hi, a.use_nan_to_num, a.nan_value, a.decimals = 1, 0.0, 100.0, 1, 0.0, 1
    a.t2, a.device = t2.data_ptr(), local_rank
    stream = torch.cuda.current_stream(device)
    a.stream = stream.cuda_stream
    for _ in range(3):
        L.check(lib.qmri_dess_t2_device(ctypes.byref(a)))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    ev0.record(stream)
    reps = 20
    for _ in range(reps):
        L.check(lib.qmri_dess_t2_device(ctypes.byref(a)))
    ev1.record(stream)
    barrier()
    ms = ev0.elapsed_time(ev1) / reps
    gbs = 16.0 * n / (ms * 1e-3) / 1e9
    return {"metric": "qDESS analytic T2 map, 384x384x160, fp32 echoes -> fp64 map", "voxels_per_s": n / (ms * 1e-3),
            "kernel_ms": ms, "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_voxel": 16}}


UNET_HW = 384
UNET_SLICES = 160
UNET_GFLOP_PER_SLICE = 70.79   # 35.39 GMAC per 384x384 slice
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak


def bench_unet(L, torch, dist, device, local_rank, world, args, barrier, red_device):
    """UNet2D slices/s (BASELINE.json configs[3]: IWOAIOAIUnet2DNormalized, 384x384x160, bf16 MFMA conv).

    A step = one whole volume (160 sagittal slices incl. whole-volume whitening) per GPU through the
    network, input resident in HBM, logits + masks written to HBM.  Random He-initialised weights of
    the reference architecture (the trained .h5 is not distributed; throughput does not depend on
    the values) and random-normal input (not zeros: DVFS, cdna_hip_programming.md rule 25)."""
    from dosma_amd.models import weights as W

    eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), UNET_HW, UNET_HW,
                         max_batch=args.unet_batch, precision="bf16", device=local_rank)
    gen = torch.Generator(device=device).manual_seed(7 + local_rank)
    x = torch.randn((UNET_SLICES, UNET_HW, UNET_HW), device=device, generator=gen) * 150 + 300
    logits = torch.empty((UNET_SLICES, UNET_HW, UNET_HW, 4), device=device)
    mask = torch.empty((UNET_SLICES, UNET_HW, UNET_HW, 4), device=device, dtype=torch.uint8)
    stream = torch.cuda.current_stream(device)
    steps = max(2, args.steps // 4)
    res = {}
    for prec in ("bf16x3", "bf16"):
        eng.set_precision(prec)
        for _ in range(max(1, args.warmup // 2)):
            eng.forward_device(x.data_ptr(), UNET_SLICES, logits.data_ptr(), mask.data_ptr(), whiten=True,
                               stream=stream.cuda_stream)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.forward_device(x.data_ptr(), UNET_SLICES, logits.data_ptr(), mask.data_ptr(), whiten=True,
                               stream=stream.cuda_stream)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=red_device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t[0].item()
        res[prec] = UNET_SLICES * world * steps / el
    eng.close()
    tf = res["bf16"] / world * UNET_GFLOP_PER_SLICE / 1e3
    return {
        "metric": "UNet2D slices/sec (IWOAIOAIUnet2DNormalized, 384x384x160, bf16 MFMA conv)",
        "value": res["bf16"], "unit": "slices/s", "steps": steps, "batch": args.unet_batch,
        "precision": "plain-bf16 mode: bf16 weights and activations (in HBM too), fp32 accumulate; the split-bf16x3 parity mode is reported beside it",
        "slices_per_s_bf16x3": res["bf16x3"],
        "data": "synthetic (random He weights of the reference architecture, random-normal input)",
        "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / MFMA_BF16_PEAK_TFLOPS, "gflop_per_slice": UNET_GFLOP_PER_SLICE,
                     "traffic": None},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--recipes", default="B,A", help="which recipes to time (A = headline, last)")
    ap.add_argument("--no-unet", action="store_true", help="skip the UNet2D slices/s leg")
    ap.add_argument("--unet-batch", type=int, default=160,
                    help="slices per pass through the network (default: the whole 160-slice volume)")
    args = ap.parse_args()

    from dosma_amd import _lib as L

    lib = L.load()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # QMRI_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices); the real runs use nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("QMRI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    red_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    local_rank = dev_index

    y = make_volume(torch, device, 20260928 + rank)
    n = y.shape[1]
    popt = torch.empty((n, 2), dtype=torch.float32, device=device)
    r2 = torch.empty(n, dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream(device)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    results = {}
    for recipe in args.recipes.split(","):  # A last: it is the headline
        a = make_args(L, y, popt, r2, stream.cuda_stream, recipe)
        a.device = local_rank
        for _ in range(args.warmup):
            L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
        ev1.record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
        if world > 1:
            t = torch.tensor([elapsed, kernel_ms], device=red_device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, kernel_ms = t[0].item(), t[1].item()
        results[recipe] = dict(elapsed=elapsed, kernel_ms=kernel_ms,
                               kernel=lib.qmri_monoexp_kernel_name(ctypes.byref(a)).decode())

    roi_run = bench_t1rho_roi(L, lib, torch, device, local_rank, args) if rank == 0 else None
    dess = bench_dess(L, lib, torch, device, local_rank, world, args, barrier) if rank == 0 or world > 1 else None
    unet = None
    if not args.no_unet:
        unet = bench_unet(L, torch, dist, device, local_rank, world, args, barrier, red_device)

    if rank == 0:
        ra = results["A"]
        total_voxels = n * world * args.steps
        value = total_voxels / ra["elapsed"]
        achieved = BYTES_PER_VOXEL * n / (ra["kernel_ms"] * 1e-3) / 1e9
        traffic = None
        valu_lane_instr = None
        prof = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(prof):
            with open(prof) as f:
                pj = json.load(f)
            traffic = pj.get("hbm_bytes_per_launch")
            valu_lane_instr = pj.get("valu_lane_instr_per_launch")
        # the bound that actually binds: vector-ALU issue.  Peak = 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz lane-
        # instructions/s (an fp64 FMA on every lane every clock = the 78.6 TFLOP/s vector fp64 figure); achieved = the
        # kernel's active lane-instructions per launch (rocprofv3 PMC, SQ_INSTS_VALU x lanes active per instruction,
        # profiles/r01e_counters.json) over the launch duration measured here
        valu_peak = 256 * 4 * 16 * 2.4e9
        valu_rate = valu_lane_instr / (ra["kernel_ms"] * 1e-3) if valu_lane_instr else None
        out = {
            "metric": "voxel-fits/sec (8-echo monoexp, 512x512x160) [+ UNet2D slices/sec under \"unet2d\"]",
            "value": value,
            "unit": "voxel-fits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ra["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "T2 monoexponential fit, 512x512x160 x 8 echoes, fp32 samples "
                            "(BASELINE.json configs[1]); MonoExponentialFit() defaults; "
                            "one volume per GPU per step",
                "voxels_per_gpu_per_step": n,
                "echoes": E,
                "parallelism": f"volumes sharded over {world} GPU(s), no data-path collective",
                "kernel": ra["kernel"],
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_voxel": BYTES_PER_VOXEL,
                "kernel_ms": ra["kernel_ms"],
                "note": "the kernel is fp64-VALU bound, not HBM bound: MINPACK's early-stopped trajectory is ~20 LM rounds "
                        "per voxel (53 charged model evaluations) of ~1050 fp64 VALU instructions each (lmpar, model "
                        "evaluation, ratio tests, forward-difference Jacobian + QR); see `valu` and DESIGN.md 3.1",
                # measured with rocprofv3 PMC on this kernel and workload (profiles/r01e_counters.json; constants, not
                # re-measured by this run): VALU pipes busy 78 % of the kernel's cycles, 40.4 of 64 lanes active per
                # VALU instruction (divergent lmpar iteration counts / rejected steps), HBM traffic 1.15x algorithmic
                "valu": {"busy_frac": 0.79, "lanes_active_frac": 0.630, "hbm_traffic_over_algorithmic": 1.15,
                         "valu_instructions_per_wave_round": 1051, "source": "profiles/r01e_counters.json",
                         "lane_instr_per_s": valu_rate, "peak_lane_instr_per_s": valu_peak,
                         "frac_of_valu_peak": (valu_rate / valu_peak) if valu_rate else None},
            },
            "runs": {
                "A_defaults_fixed_p0": {"voxel_fits_per_s": n * world * args.steps / ra["elapsed"],
                                        "kernel_ms": ra["kernel_ms"]},
            },
        }
        if unet is not None:
            out["unet2d"] = unet
        if dess is not None:
            out["dess_t2"] = dess
        if roi_run is not None:
            out["runs"]["cfg2_t1rho_roi"] = roi_run
        if "B" in results:
            out["runs"]["B_polyfit_init"] = {
                "voxel_fits_per_s": n * world * args.steps / results["B"]["elapsed"],
                "kernel_ms": results["B"]["kernel_ms"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(y)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
