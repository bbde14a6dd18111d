"""Where conv_c4_kernel's work items spend their time (experiment build: python -m dosma_amd.build --variant tl -DQMRI_C4_TIMELINE;
run with DOSMA_AMD_LIB=.../libqmri_hip_tl.so): block 8 of every launch timestamps its second and third item (s_memrealtime, 10 ns)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dosma_amd import _lib as L
from dosma_amd.models import weights as W

slices = int(sys.argv[1]) if len(sys.argv) > 1 else 160
eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), 384, 384, max_batch=slices, precision="fp16x3")
dev = torch.device("cuda", 0)
x = torch.randn((slices, 384, 384), device=dev)
logits = torch.empty((slices, 384, 384, 4), device=dev)
mask = torch.empty((slices, 384, 384, 4), device=dev, dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
lib = ctypes.CDLL(os.environ["DOSMA_AMD_LIB"])
buf = (ctypes.c_ulonglong * (256 * 16))()
for rep in range(2):
    eng.forward_device(x.data_ptr(), slices, logits.data_ptr(), mask.data_ptr(), whiten=True, stream=st)
    torch.cuda.synchronize()
    n = lib.qmri_debug_c4_timeline(buf, 256)
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16)[:n].astype(np.int64)
print("per launch (the forward's c4 layers in order), microseconds; item 2 | item 3: k loop, epilogue, next_item")
for r in a:
    ct, flat, chunks, items = r[:4]
    print(f"CT {ct} flat {flat} chunks {chunks:2d} items {items:3d} | " + " ".join(f"{v / 100:7.2f}" for v in r[4:7]) + " | " + " ".join(f"{v / 100:7.2f}" for v in r[9:12])
          + (f"  first 3 steps {r[7] / 100:.2f} / {r[12] / 100:.2f}, their wait + barrier {r[8] / 100:.2f} / {r[13] / 100:.2f}" if r[7] else "")
          + f"   per step {(r[4] + r[9]) / 2 / (18 * chunks) / 100:.3f} us, epilogue + next_item = {(r[5] + r[6] + r[10] + r[11]) / 2 / 100:.2f} us = {(r[5] + r[6] + r[10] + r[11]) / (r[4] + r[9] + r[5] + r[6] + r[10] + r[11]) * 100:.1f} % of the item")
