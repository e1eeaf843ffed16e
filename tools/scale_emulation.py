#!/usr/bin/env python3
"""
What each rank of an N-GPU run would do, timed on ONE GPU: for N = 1, 2, 4, 8
every rank's row shard of the workload (as bench.py --gpus N builds it: the
shard in its PACKED column space, ``auto_schedule`` on its rows, the kernel
reading the packed buffer ``X[ucols]``; source cells numbered as an MPAS mesh
numbers them) is launched back to back and timed with one HIP event pair; the slowest rank gives the kernel-phase time
of the N-GPU step.  No collective is involved in the timed region of
bench.py either (X is broadcast before it), so this is the kernel-phase
projection, not a measurement of xGMI.

    python tools/scale_emulation.py [--workload config3] > profiles/..json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config3')
    ap.add_argument('--reps', type=int, default=100)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    K = synthetic.CONFIGS[args.workload]['K']
    m = synthetic.make_config(args.workload, device=dev, locality='mesh')
    full = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    xs = [torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
          for _ in range(2)]
    out = {'workload': args.workload, 'K': K, 'n_b': m.n_b, 'ranks': {}}
    base = None
    for world in (1, 2, 4, 8):
        times, packed_rows, gather_ms = [], [], []
        for rank in range(world):
            if world > 1:
                plan, ucols = full.shard(rank, world).packed()
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record()
                xp = [engine.gather_rows(x, 0, ucols) for x in xs]
                b.record()
                torch.cuda.synchronize()
                gather_ms.append(a.elapsed_time(b) / len(xs))
                packed_rows.append(int(ucols.shape[0]))
            else:
                plan, xp = full, xs
                packed_rows.append(m.n_a)
            plan.auto_schedule(m.dst_dims)
            ys = [torch.empty((plan.n_b, K), device=dev, dtype=torch.float64)
                  for _ in range(2)]

            def launch(i):
                engine.remap_tensor(plan, m.dst_dims if world == 1 else None,
                                    xp[i % 2], [0], engine.MODE_FRACB,
                                    out=ys[i % 2].reshape(
                                        m.dst_dims + (K,)) if world == 1
                                    else ys[i % 2])
            for i in range(10):
                launch(i)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(args.reps):
                launch(i)
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / args.reps)
        slow = max(times)
        if world == 1:
            base = slow
        out['ranks'][world] = {
            'ms_per_rank': [round(t, 5) for t in times],
            'ms_step': round(slow, 5),
            'packed_rows_fraction_of_broadcast': round(
                sum(packed_rows) / (world * m.n_a), 4),
            'local_gather_ms_max': round(max(gather_ms), 5) if gather_ms
            else 0.0,
            'cell_fields_per_s': m.n_b * K / (slow * 1e-3),
            'speedup': round(base / slow, 3),
            'efficiency': round(base / slow / world, 3)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
