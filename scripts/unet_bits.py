"""SHA-1 of the parity-mode logits and masks of a seeded forward pass (to compare two builds of the library bit for bit:
DOSMA_AMD_LIB=... python scripts/unet_bits.py [hw] [slices])."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dosma_amd import _lib as L
from dosma_amd.models import weights as W

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 384
slices = int(sys.argv[2]) if len(sys.argv) > 2 else 24
eng = L.Unet2dEngine(W.to_abi_order(W.random_weights(seed=0)), hw, hw, max_batch=slices, precision="fp16x3")
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
x = torch.randn((slices, hw, hw), device=dev, generator=g) * 3 + 1
logits = torch.empty((slices, hw, hw, 4), device=dev)
mask = torch.empty((slices, hw, hw, 4), device=dev, dtype=torch.uint8)
eng.forward_device(x.data_ptr(), slices, logits.data_ptr(), mask.data_ptr(), whiten=True, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(f"{hw}^2 x {slices}: logits sha1 {hashlib.sha1(logits.cpu().numpy().tobytes()).hexdigest()}  masks sha1 {hashlib.sha1(mask.cpu().numpy().tobytes()).hexdigest()}")
