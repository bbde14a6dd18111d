"""End-to-end time of IWOAIOAIUnet2DNormalized.generate_mask on a 384 x 384 x 160 MedicalVolume (host in, host out)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dosma_amd as dm
from dosma_amd.models import IWOAIOAIUnet2DNormalized
from oracle import unet_oracle as uo  # weights generator only

w = uo.make_weights(seed=3)
H, W, S = 384, 384, 160
rng = np.random.default_rng(0)
vol = (rng.standard_normal((H, W, S)) * 80 + 200).astype(np.float32)
aff = np.array([[0, 0, 1.5, 0.0], [0, -0.4, 0, 0.0], [-0.4, 0, 0, 0.0], [0, 0, 0, 1.0]])  # sagittal already
mv = dm.MedicalVolume(vol, aff)
import itertools
for precision, gb in itertools.product(("fp16x3", "bf16"), (64, 160)):
    IWOAIOAIUnet2DNormalized.precision = precision
    IWOAIOAIUnet2DNormalized.gpu_batch = gb
    model = IWOAIOAIUnet2DNormalized((H, W, 1), w, force_weights=True)
    for rep in range(3):
        t0 = time.perf_counter()
        out = model.generate_mask(mv)
        dt = time.perf_counter() - t0
    print(f"generate_mask [{precision}, gpu_batch {gb}]: {dt*1e3:.1f} ms  -> {S/dt:.0f} slices/s end to end "
          f"(numpy volume in, 4 numpy masks out)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); model.generate_mask(mv); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
