"""Can this platform move host->device and device->host at the same time?  Pinned and pageable, torch copies on two streams
(and two threads for the pageable case: a pageable copy blocks its caller)."""
import threading, time
import torch

dev = torch.device("cuda", 0)
n_up, n_dn = 1342177280 // 4, 671088640 // 4
up_d = torch.empty(n_up, dtype=torch.float32, device=dev)
dn_d = torch.ones(n_dn, dtype=torch.float32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(tag, up_h, dn_h, threads):
    def up():
        with torch.cuda.stream(s1):
            up_d.copy_(up_h, non_blocking=True)
            s1.synchronize()

    def dn():
        with torch.cuda.stream(s2):
            dn_h.copy_(dn_d, non_blocking=True)
            s2.synchronize()

    for what in ("up", "down", "both"):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            if what == "up":
                up()
            elif what == "down":
                dn()
            elif threads:
                a, b = threading.Thread(target=up), threading.Thread(target=dn)
                a.start(); b.start(); a.join(); b.join()
            else:
                with torch.cuda.stream(s1):
                    up_d.copy_(up_h, non_blocking=True)
                with torch.cuda.stream(s2):
                    dn_h.copy_(dn_d, non_blocking=True)
                s1.synchronize(); s2.synchronize()
            best = min(best, time.perf_counter() - t)
        gb = {"up": n_up * 4, "down": n_dn * 4, "both": (n_up + n_dn) * 4}[what] / 1e9
        print(f"{tag:9s} {what:5s} {best * 1e3:7.2f} ms  {gb / best:6.1f} GB/s")


run("pinned", torch.empty(n_up, dtype=torch.float32).pin_memory(), torch.empty(n_dn, dtype=torch.float32).pin_memory(), False)
up_p = torch.ones(n_up, dtype=torch.float32); dn_p = torch.zeros(n_dn, dtype=torch.float32)
run("pageable", up_p, dn_p, True)
