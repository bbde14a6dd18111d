"""Lane-utilisation counters of the fit kernel per phase (debug build with -DQMRI_STATS, built on the spot).
Prints, for recipes A and B on the bench volume: rounds, busy-lane fraction, lmpar / QR participation."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dosma_amd import build as B

so = os.path.join(ROOT, "gpurun_out", "libqmri_stats.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
src = [os.path.join(ROOT, "dosma_amd", "csrc", f) for f in B.SOURCES]
cmd = [B._hipcc(), f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-DQMRI_STATS",
       "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "dosma_amd", "csrc"), "-o", so] + src
subprocess.check_call(cmd)
from dosma_amd import _lib as L
L._SO = so
import torch
import bench
lib = L.load()
lib.qmri_debug_fit_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device("cuda", 0)
n = int(os.environ.get("N", 1 << 23))
y = bench.make_volume(torch, dev, 20260928)[:, :n].contiguous()
popt = torch.empty((n, 2), dtype=torch.float32, device=dev)
r2 = torch.empty(n, dtype=torch.float32, device=dev)
for recipe in ("A", "B"):
    a = bench.make_args(L, y, popt, r2, torch.cuda.current_stream().cuda_stream, recipe)
    buf = (ctypes.c_ulonglong * 16)()
    lib.qmri_debug_fit_stats(buf, 1)
    L.check(lib.qmri_monoexp_fit_device(ctypes.byref(a), None))
    torch.cuda.synchronize()
    lib.qmri_debug_fit_stats(buf, 1)
    rounds, busy, lm_in, lm_lane, lm_wave, qr, fin, _ = [int(v) for v in buf][:8]
    cyc = [int(v) for v in buf][8:15]
    tot = max(sum(cyc), 1)
    print('   s_memtime share per section: ' + '  '.join(f'{n} {c/tot:.3f}' for n, c in zip(('refill+epilogue', 'lmpar-setup', 'lmpar-loop', 'model-eval', 'accept-logic', 'fd-jacobian', 'qr'), cyc)) + f'   cycles/round {tot/rounds:.0f}')
    slots = rounds * 64
    print(f"recipe {recipe}: rounds {rounds}  busy/slot {busy/slots:.3f}  lmpar-entering/slot {lm_in/slots:.3f}  "
          f"QR/slot {qr/slots:.3f}  finished {fin}  rounds/fit {busy/max(fin,1):.2f}")
    print(f"   lmpar loop: lane-iterations {lm_lane}  wave-iterations {lm_wave}  "
          f"=> lanes active per lmpar wave-iteration {lm_lane/max(lm_wave,1)/64:.3f}; "
          f"mean iterations per entering lane {lm_lane/max(lm_in,1):.2f}, per wave-round {lm_wave/rounds:.2f}")
