#!/usr/bin/env python3
"""Host <-> device copy rates on this box (pageable / pinned) and host memcpy."""
import time

import numpy as np
import torch

n = 235160 * 512
a = np.random.default_rng(0).standard_normal(n)
dev = torch.device('cuda', 0)
t = torch.from_numpy(a)
d = torch.empty(n, dtype=torch.float64, device=dev)
pin = torch.empty(n, dtype=torch.float64).pin_memory()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


gb = n * 8 / 1e9
print(f'{gb:.2f} GB')
print('H2D pageable  %.1f GB/s' % (gb / timed(lambda: d.copy_(t))))
print('H2D pinned    %.1f GB/s' % (gb / timed(lambda: d.copy_(pin, non_blocking=True))))
print('host memcpy   %.1f GB/s' % (gb / timed(lambda: pin.copy_(t))))
out = torch.empty(n, dtype=torch.float64)
print('D2H pageable  %.1f GB/s' % (gb / timed(lambda: out.copy_(d))))
print('D2H pinned    %.1f GB/s' % (gb / timed(lambda: pin.copy_(d, non_blocking=True))))
print('D2H .cpu()    %.1f GB/s' % (gb / timed(lambda: d.cpu())))
t0 = time.perf_counter()
p2 = torch.empty(n, dtype=torch.float64).pin_memory()
print('pin_memory alloc of %.2f GB: %.3f s' % (gb, time.perf_counter() - t0))
torch.set_num_threads(8)
print('host memcpy 8 threads %.1f GB/s' % (gb / timed(lambda: pin.copy_(t))))
