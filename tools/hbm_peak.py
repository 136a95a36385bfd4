#!/usr/bin/env python3
"""Achievable HBM bandwidth on this GPU: device-to-device copy and fill of 4 GiB (torch)."""
import time
import torch
n = 1 << 30
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn, nbytes in (("copy (read+write)", lambda: y.copy_(x), 8 * n), ("fill (write)", lambda: x.fill_(1.0), 4 * n),
                         ("sum (read)", lambda: x.sum(), 4 * n)):
    fn(); torch.cuda.synchronize()
    best = 0
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = max(best, nbytes / dt / 1e9)
    print(f"{name:18s} {best:8.0f} GB/s")
