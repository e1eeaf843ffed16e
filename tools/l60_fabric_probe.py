#!/usr/bin/env python3
"""(8, nCells, 60) against (8, nCells, 64) on config 3's map with the fabric
taken away piece by piece (diagnostic build, tune[6]: 1 = no Y stores, 2 = X
from the first 1 024 rows -- L2 hits --, 3 = both): where do the 12 % that 60
levels cost over 64 go?  GPU box only.

    REMAP_HIP_LIB=tools/_build/libremap_hip_diag.so python tools/l60_fabric_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    m = synthetic.make_config('config3', device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    plan.auto_schedule(m.dst_dims)
    for L in (60, 64):
        xs = [torch.randn((8, m.n_a, L), device=dev, dtype=torch.float64)
              for _ in range(3)]
        ys = [torch.empty((8,) + tuple(m.dst_dims) + (L,), device=dev,
                          dtype=torch.float64) for _ in range(3)]
        for tag, d in (('all', 0), ('no stores', 1), ('X from L2', 2),
                       ('neither', 3)):
            tune = [10, 0, 0, 1, 0, 0, d]

            def run(i):
                engine.remap_tensor(plan, m.dst_dims, xs[i % 3], [1],
                                    engine.MODE_FRACB, tune=tune,
                                    out=ys[i % 3])
            for i in range(5):
                run(i)
            best = []
            for _ in range(3):
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(30):
                    run(i)
                b.record()
                torch.cuda.synchronize()
                best.append(a.elapsed_time(b) / 30)
            print(f'L={L} {tag:10s} {min(best):.4f} ms')


if __name__ == '__main__':
    main()
