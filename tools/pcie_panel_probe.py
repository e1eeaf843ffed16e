#!/usr/bin/env python3
"""
Column panels over PCIe (GPU box): how fast does a (n_a, K) host array go up
-- and a (n_b, K) result come down -- in column panels of `kp` columns,
against the whole array in one contiguous copy?  Decides the form of
host_path's column-panel pipeline.

    python tools/pcie_panel_probe.py [n_a K kp]
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

n_a = int(sys.argv[1]) if len(sys.argv) > 1 else 235160
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
kp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device('cuda', 0)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib',
                               'libamdhip64.so'))
hip.hipMemcpy2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.c_size_t, ctypes.c_size_t,
                                 ctypes.c_int, ctypes.c_void_p]
H2D, D2H = 1, 2
x = np.random.default_rng(0).standard_normal((n_a, K))
xt = torch.from_numpy(x)
x_d = torch.empty((n_a, K), dtype=torch.float64, device=dev)
panel_d = torch.empty((n_a, kp), dtype=torch.float64, device=dev)
out_h = torch.empty((n_a, K), dtype=torch.float64).pin_memory()
pin_panel = torch.empty((n_a, kp), dtype=torch.float64).pin_memory()
stream = torch.cuda.current_stream(dev)
sp = ctypes.c_void_p(stream.cuda_stream)
GB = x.nbytes / 1e9


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


t = timed(lambda: x_d.copy_(xt, non_blocking=True))
print(f'H2D whole, pageable, contiguous: {t * 1e3:.1f} ms  {GB / t:.1f} GB/s')


def h2d_2d_pageable():
    for p0 in range(0, K, kp):
        rc = hip.hipMemcpy2DAsync(
            panel_d.data_ptr(), kp * 8, x.ctypes.data + p0 * 8, K * 8,
            kp * 8, n_a, H2D, sp)
        assert rc == 0, rc


t = timed(h2d_2d_pageable)
print(f'H2D panels of {kp}, hipMemcpy2DAsync from pageable: '
      f'{t * 1e3:.1f} ms  {GB / t:.1f} GB/s')


def h2d_torch_strided():
    for p0 in range(0, K, kp):
        panel_d.copy_(xt[:, p0:p0 + kp], non_blocking=True)


t = timed(h2d_torch_strided)
print(f'H2D panels, torch strided copy_: {t * 1e3:.1f} ms  {GB / t:.1f} GB/s')


def h2d_staged():
    for p0 in range(0, K, kp):
        np.copyto(pin_panel.numpy(), x[:, p0:p0 + kp])
        panel_d.copy_(pin_panel, non_blocking=True)
        torch.cuda.synchronize()


t = timed(h2d_staged)
print(f'H2D panels, numpy -> pinned panel -> device (1 thread, serial): '
      f'{t * 1e3:.1f} ms  {GB / t:.1f} GB/s')
xp = torch.empty((n_a, K), dtype=torch.float64).pin_memory()
xp.copy_(xt)


def h2d_2d_pinned():
    for p0 in range(0, K, kp):
        rc = hip.hipMemcpy2DAsync(
            panel_d.data_ptr(), kp * 8, xp.data_ptr() + p0 * 8, K * 8,
            kp * 8, n_a, H2D, sp)
        assert rc == 0, rc


t = timed(h2d_2d_pinned)
print(f'H2D panels, hipMemcpy2DAsync from PINNED: {t * 1e3:.1f} ms  '
      f'{GB / t:.1f} GB/s')
t = timed(lambda: out_h.copy_(x_d, non_blocking=True))
print(f'D2H whole into pinned: {t * 1e3:.1f} ms  {GB / t:.1f} GB/s')


def d2h_2d():
    for p0 in range(0, K, kp):
        rc = hip.hipMemcpy2DAsync(
            out_h.data_ptr() + p0 * 8, K * 8, panel_d.data_ptr(), kp * 8,
            kp * 8, n_a, D2H, sp)
        assert rc == 0, rc


t = timed(d2h_2d)
print(f'D2H panels of {kp} into pinned (strided host side): {t * 1e3:.1f} ms'
      f'  {GB / t:.1f} GB/s')


def d2h_torch():
    for p0 in range(0, K, kp):
        out_h[:, p0:p0 + kp].copy_(panel_d, non_blocking=True)


t = timed(d2h_torch)
print(f'D2H panels, torch strided copy_: {t * 1e3:.1f} ms  {GB / t:.1f} GB/s')
# device-side strided gather of a panel out of a whole uploaded array is free
# by comparison: X could also go up in ROW chunks (contiguous) -- but a
# destination row needs source rows from everywhere
up = torch.cuda.Stream(dev)
dn = torch.cuda.Stream(dev)


def duplex():
    with torch.cuda.stream(up):
        x_d.copy_(xt, non_blocking=True)
    with torch.cuda.stream(dn):
        out_h.copy_(x_d, non_blocking=True)


t = timed(duplex)
print(f'whole up + whole down at once (two streams): {t * 1e3:.1f} ms')
