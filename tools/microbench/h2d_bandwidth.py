import torch, time
n = 576_000_000
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for chunk in (n, n // 32, n // 256):
    torch.cuda.synchronize(); t = time.perf_counter()
    for r in range(3):
        for o in range(0, n, chunk):
            d[o:o + chunk].copy_(h[o:o + chunk], non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f"H2D pinned, chunk {chunk/1e6:.1f} MB: {n/dt/1e9:.1f} GB/s")
import numpy as np
a = np.ones(n, np.uint8); b = np.empty(n, np.uint8)
t = time.perf_counter(); b[:] = a; dt = time.perf_counter() - t
print(f"single-thread host memcpy: {n/dt/1e9:.1f} GB/s")
