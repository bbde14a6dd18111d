import os, sys, time, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np
from dosma_amd import _lib as L
from oracle import unet_oracle as uo
from dosma_amd.models import weights as W
import torch
w = uo.make_weights(seed=3)
eng = L.Unet2dEngine(W.to_abi_order(w), 384, 384, n_classes=4, max_batch=160, precision="fp16x3")
vol = (np.random.default_rng(0).standard_normal((384, 384, 160)) * 80 + 200).astype(np.float32)
for rep in range(4):
    t0 = time.perf_counter()
    planes = eng.segment_volume(vol, whiten=True, eps=0.0)
    t1 = time.perf_counter()
    print(f"segment_volume total {1e3*(t1-t0):.1f} ms")
# pieces
x = torch.empty(384*384*160, dtype=torch.float32, device="cuda")
hv = torch.from_numpy(vol.reshape(-1))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); x.copy_(hv); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"  H2D 94 MB pageable (torch): {1e3*(t1-t0):.2f} ms")
m = torch.empty(4*384*384*160, dtype=torch.uint8, device="cuda")
for rep in range(3):
    out = np.empty(4*384*384*160, np.uint8)
    ho = torch.from_numpy(out)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ho.copy_(m); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"  D2H 94 MB into a fresh numpy array (torch): {1e3*(t1-t0):.2f} ms")
out = np.empty(4*384*384*160, np.uint8); ho = torch.from_numpy(out)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ho.copy_(m); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"  D2H 94 MB into a reused numpy array (torch): {1e3*(t1-t0):.2f} ms")
t0 = time.perf_counter(); o2 = np.empty(4*384*384*160, np.uint8); o2[::4096] = 0; t1 = time.perf_counter()
print(f"  touching 94 MB of fresh pages: {1e3*(t1-t0):.2f} ms")
