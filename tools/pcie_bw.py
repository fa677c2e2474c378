import torch, time
n = 256 * 1024 * 1024 // 8
d = torch.zeros(n, dtype=torch.float64, device="cuda")
h = torch.empty(n, dtype=torch.float64, pin_memory=True)
hp = torch.empty(n, dtype=torch.float64)
for name, dst in (("pinned", h), ("pageable", hp)):
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        dst.copy_(d, non_blocking=False); torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print("D2H %s: %.1f GB/s" % (name, n * 8 / dt / 1e9))
for name, src in (("pinned", h), ("pageable", hp)):
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        d.copy_(src, non_blocking=False); torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print("H2D %s: %.1f GB/s" % (name, n * 8 / dt / 1e9))
import numpy as np
a = np.zeros(n); 
t = time.perf_counter(); b = a.copy(); print("host memcpy 1 thread: %.1f GB/s" % (n*8/(time.perf_counter()-t)/1e9))
