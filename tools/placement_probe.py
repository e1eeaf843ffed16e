#!/usr/bin/env python3
"""
Does the kernel's time depend on WHERE X and Y lie in HBM relative to one
another?  (GPU box only.)  bench.py's three rotated buffer sets differ by
4-6 % in a fixed pattern; this probe carves X and Y out of one arena at
controlled offsets and times the metric launch for each.

    python tools/placement_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    m = synthetic.make_config('config3', device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    plan.auto_schedule(m.dst_dims)
    K = 512
    xb, yb = m.n_a * K * 8, m.n_b * K * 8
    arena = torch.empty(xb + yb + (1 << 30), dtype=torch.uint8, device=dev)
    base = arena.data_ptr()
    print(f'arena at {base:#x} (mod 2 MiB {base % (2 << 20):#x}), X {xb} B, '
          f'Y {yb} B')
    x = arena[:xb].view(torch.float64).reshape(m.n_a, K)
    x.normal_()
    x_end = (xb + 4095) // 4096 * 4096

    def timed(y, reps=40):
        for _ in range(10):
            engine.apply_strided(plan, x, y, n_batch=1, k_inner=K,
                                 x_row_stride=K, x_batch_stride=0,
                                 y_row_stride=K, y_batch_stride=0,
                                 mode=engine.MODE_FRACB)
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            engine.apply_strided(plan, x, y, n_batch=1, k_inner=K,
                                 x_row_stride=K, x_batch_stride=0,
                                 y_row_stride=K, y_batch_stride=0,
                                 mode=engine.MODE_FRACB)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    # warm the clocks
    y0 = arena[x_end:x_end + yb].view(torch.float64).reshape(m.n_b, K)
    for _ in range(3):
        timed(y0)
    offs = [0, 4 << 10, 8 << 10, 16 << 10, 32 << 10, 64 << 10, 128 << 10,
            256 << 10, 512 << 10, 1 << 20, 2 << 20, 4 << 20, 8 << 20,
            16 << 20, 32 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20,
            (1 << 20) + 4096, (3 << 20), (5 << 20) + (64 << 10)]
    for rnd in range(2):
        for d in offs:
            y = arena[x_end + d:x_end + d + yb].view(torch.float64).reshape(
                m.n_b, K)
            t = timed(y)
            print(f'round {rnd} Y - X_end = {d:>10d} B  ((Y - X) mod 1 MiB = '
                  f'{(x_end + d) % (1 << 20):>8d}): {t:.4f} ms')
    # separately allocated tensors, as bench.py has them
    for i in range(4):
        xs = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
        ys = torch.empty((m.n_b, K), device=dev, dtype=torch.float64)
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            engine.remap_tensor(plan, m.dst_dims, xs, [0], engine.MODE_FRACB,
                                out=ys.view(m.dst_dims + (K,)))
        a.record()
        for _ in range(40):
            engine.remap_tensor(plan, m.dst_dims, xs, [0], engine.MODE_FRACB,
                                out=ys.view(m.dst_dims + (K,)))
        b.record()
        torch.cuda.synchronize()
        print(f'separate tensors {i}: X {xs.data_ptr():#x} Y '
              f'{ys.data_ptr():#x}  {a.elapsed_time(b) / 40:.4f} ms')
        keep = (xs, ys) if i % 2 == 0 else None   # vary what stays allocated
        del xs, ys


if __name__ == '__main__':
    main()
