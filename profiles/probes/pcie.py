"""PCIe rates of the box: pinned host <-> device, one direction at a time and both at once (two streams)."""
import time
import torch

n = 512 << 20
h_up = torch.empty(n, dtype=torch.uint8).pin_memory()
h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def up():
    with torch.cuda.stream(s1):
        d_a.copy_(h_up, non_blocking=True)


def dn():
    with torch.cuda.stream(s2):
        h_dn.copy_(d_b, non_blocking=True)


def both():
    up(); dn()


print("H2D %.1f GB/s   D2H %.1f GB/s   both at once %.1f + %.1f GB/s" % (n / t(up) / 1e9, n / t(dn) / 1e9, n / t(both) / 1e9, n / t(both) / 1e9))
# 4 MB pieces (one column each)
pieces = 128
def up_pieces():
    with torch.cuda.stream(s1):
        for i in range(pieces):
            d_a[i * (4 << 20):(i + 1) * (4 << 20)].copy_(h_up[i * (4 << 20):(i + 1) * (4 << 20)], non_blocking=True)
print("H2D in 4 MB pieces %.1f GB/s" % (pieces * (4 << 20) / t(up_pieces) / 1e9))
