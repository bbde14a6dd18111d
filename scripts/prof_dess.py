"""The qDESS T2 kernel on a batch of 8 volumes (3 GB of algorithmic traffic per launch) for rocprofv3: scripts/pmc_dess.sh."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dosma_amd import _lib as L

lib = L.load()
dev = torch.device("cuda", 0)
n = 8 * 384 * 384 * 160
gen = torch.Generator(device=dev).manual_seed(11)
e1 = torch.rand(n, device=dev, generator=gen) * 780 + 20
e2 = e1 * (torch.rand(n, device=dev, generator=gen) * 0.88 + 0.02)
t2 = torch.empty(n, device=dev, dtype=torch.float64)
a = L.QmriDessArgs()
a.echo1, a.echo2, a.dtype, a.out_dtype, a.N = e1.data_ptr(), e2.data_ptr(), L.QMRI_F32, L.QMRI_F64, n
a.c0, a.k, a.c1 = -27.864, 0.0434, 3.9e-3
a.use_bounds, a.lo, a.hi, a.use_nan_to_num, a.nan_value, a.decimals = 1, 0.0, 100.0, 1, 0.0, 1
a.t2, a.device = t2.data_ptr(), 0
a.stream = torch.cuda.current_stream(dev).cuda_stream
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    L.check(lib.qmri_dess_t2_device(ctypes.byref(a)))
torch.cuda.synchronize()
print("ok", float(t2[:1000].mean()))
