#!/usr/bin/env python3
"""
Does WHERE the field and the result lie in HBM matter?  bench.py's three
rotated buffer sets sometimes differ by 5-7 % in launch time inside one
process.  X and Y are placed at chosen offsets inside one big allocation and
the metric launch (config 3, K = 512) is timed per placement (GPU box only).

    python tools/placement_sweep.py [--reps 30]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--workload', default='config3')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    K = synthetic.CONFIGS[args.workload]['K']
    m = synthetic.make_config(args.workload, device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    plan.auto_schedule(m.dst_dims)
    xb, yb = m.n_a * K * 8, m.n_b * K * 8
    MiB = 1 << 20
    arena = torch.empty(8 << 30, dtype=torch.uint8, device=dev)
    base = arena.data_ptr()
    print(f'arena at {base:#x} (mod 2 MiB: {base % (2 * MiB):#x}); X '
          f'{xb / MiB:.1f} MiB, Y {yb / MiB:.1f} MiB')
    master = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)

    def view(off, rows):
        n = rows * K * 8
        return arena[off:off + n].view(torch.float64).view(rows, K)

    def timed(ox, oy):
        x, y = view(ox, m.n_a), view(oy, m.n_b)
        x.copy_(master)
        for _ in range(5):
            engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB,
                                out=y.view(m.dst_dims + (K,)))
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB,
                                out=y.view(m.dst_dims + (K,)))
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.reps

    def up(v, a):
        return (v + a - 1) // a * a
    y0 = up(xb, 2 * MiB)          # Y right behind X, 2 MiB aligned
    print('-- Y offset relative to a 2 MiB boundary (X at 0)')
    for d in (0, 256, 1024, 4096, 16384, 65536, 262144, MiB, MiB + 4096):
        print(f'   dy = {d:8d}: {timed(0, y0 + d) * 1e3:8.1f} us')
    print('-- X offset relative to a 2 MiB boundary (Y fixed, aligned)')
    y1 = up(xb + 4 * MiB, 2 * MiB)
    for d in (0, 256, 4096, 65536, MiB):
        print(f'   dx = {d:8d}: {timed(d, y1) * 1e3:8.1f} us')
    print('-- the same relative placement at other places of the arena')
    for shift in (0, 64, 512, 1024, 2048, 3072, 4096):
        off = shift * MiB
        if off + y0 + yb > arena.numel():
            break
        print(f'   +{shift:5d} MiB: {timed(off, off + y0) * 1e3:8.1f} us')
    print('-- separate torch allocations (what bench.py does), five pairs')
    keep = []
    for i in range(5):
        x = master.clone()
        y = torch.empty((m.n_b, K), device=dev, dtype=torch.float64)
        keep += [x, y]
        for _ in range(5):
            engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB,
                                out=y.view(m.dst_dims + (K,)))
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB,
                                out=y.view(m.dst_dims + (K,)))
        b.record()
        torch.cuda.synchronize()
        print(f'   pair {i}: X {x.data_ptr():#x} Y {y.data_ptr():#x} '
              f'(Y - X = {(y.data_ptr() - x.data_ptr()) / MiB:.2f} MiB): '
              f'{a.elapsed_time(b) / args.reps * 1e3:8.1f} us')


if __name__ == '__main__':
    main()
