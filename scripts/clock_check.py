"""Does `rocm-smi --setperfdeterminism N` hold the shader clock under the parity-mode forward?  (round 6: the c4 race hunt's fixed-clock run)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dosma_amd import _lib as L
from dosma_amd.models import weights as W
dev = torch.device("cuda", 0)
eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), 384, 384, max_batch=160, precision="fp16x3", device=0)
x = torch.randn((160, 384, 384), device=dev) * 150 + 300
lg = torch.empty((160, 384, 384, 4), device=dev); mk = torch.empty((160, 384, 384, 4), device=dev, dtype=torch.uint8)
st = torch.cuda.current_stream(dev)
sp = bench.GpuSampler(torch, 0)
for _ in range(10):
    eng.forward_device(x.data_ptr(), 160, lg.data_ptr(), mk.data_ptr(), whiten=True, stream=st.cuda_stream)
torch.cuda.synchronize()
with sp:
    t0 = time.perf_counter()
    for _ in range(60):
        eng.forward_device(x.data_ptr(), 160, lg.data_ptr(), mk.data_ptr(), whiten=True, stream=st.cuda_stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(f"{(t1 - t0) / 60 * 1e3:.2f} ms per forward", sp.summary(t0 + 0.3 * (t1 - t0), t1))
