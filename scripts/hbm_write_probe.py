"""HBM write / copy bandwidth as torch sees it (a fill kernel and a device-to-device copy of 4 GB), to put the write-heavy
UNet kernels (enc0: 3.8 GB written per forward) against a pure stream."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 30
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
t = timed(lambda: x.zero_())
print(f"fill 4 GiB: {t:.3f} ms -> {4 * n / t / 1e9:.2f} TB/s written")
t = timed(lambda: y.copy_(x))
print(f"copy 4 GiB: {t:.3f} ms -> {4 * n / t / 1e9:.2f} TB/s read + {4 * n / t / 1e9:.2f} TB/s written")
t = timed(lambda: x.sum())
print(f"sum  4 GiB: {t:.3f} ms -> {4 * n / t / 1e9:.2f} TB/s read")
