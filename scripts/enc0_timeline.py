"""Where enc0_kernel's two groups spend a half-period (experiment build: python -m dosma_amd.build --variant tl -DQMRI_ENC0_TIMELINE
[-DQMRI_C4_TIMELINE]; run with DOSMA_AMD_LIB=.../libqmri_hip_tl.so): block 8 timestamps half-periods 20 .. 23 (s_memrealtime, 10 ns).
marks: 0 start | 1 first phase done (M group: the first 8 of conv 2's 18 half-steps; other group: output + stores + pooling) |
2 after the barrier | 3 second phase done (M group: the other 10 half-steps; other group: conv 1 of its next tile) | 4 after the barrier"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dosma_amd import _lib as L
from dosma_amd.models import weights as W

slices = 160
eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), 384, 384, max_batch=slices, precision="fp16x3")
dev = torch.device("cuda", 0)
x = torch.randn((slices, 384, 384), device=dev)
logits = torch.empty((slices, 384, 384, 4), device=dev)
mask = torch.empty((slices, 384, 384, 4), device=dev, dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
lib = ctypes.CDLL(os.environ["DOSMA_AMD_LIB"])
buf = (ctypes.c_ulonglong * 64)()
for rep in range(2):
    eng.forward_device(x.data_ptr(), slices, logits.data_ptr(), mask.data_ptr(), whiten=True, stream=st)
    torch.cuda.synchronize()
assert lib.qmri_debug_enc0_timeline(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4, 2, 8).astype(np.int64)
t0 = a[0, :, 0].min()
for h in range(4):
    for g in range(2):
        r = a[h, g]
        role = "M" if (20 + h) % 2 == g else "O"
        print(f"half-period {20 + h} group {g} ({role}): start {(r[0] - t0) / 100:7.2f} us | phase 1 {(r[1] - r[0]) / 100:5.2f} | barrier {(r[2] - r[1]) / 100:5.2f} | phase 2 {(r[3] - r[2]) / 100:5.2f} | barrier {(r[4] - r[3]) / 100:5.2f}")
