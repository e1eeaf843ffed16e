#!/usr/bin/env python3
"""(8, nCells, L) on config 3's map: 2 x 2 groups (the plan's schedule)
against groups of 4 CONSECUTIVE rows (whose four 480-byte runs of Y are one
1 920-byte = 15-line block written by one wave).  GPU box only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    m = synthetic.make_config('config3', device=dev, locality='mesh')
    ref = {}
    for shape in ('2x2', '1x4', '1x8', '2x4'):
        plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                              m.n_a, m.n_b, index_base=1,
                                              device=dev)
        plan.auto_schedule(m.dst_dims)
        if shape == '1x4':
            r = plan.build_groups(None, rows=4)
        elif shape == '1x8':
            r = plan.build_groups(None, rows=8)
        elif shape == '2x4':
            r = plan.build_groups(m.dst_dims, super_tile=1 << 30, rows=8)
        else:
            r = plan.groups['union'] / plan.nnz
        for L in (60, 64, 61, 100):
            xs = [torch.randn((8, m.n_a, L), device=dev, dtype=torch.float64)
                  for _ in range(3)]
            ys = [torch.empty((8,) + tuple(m.dst_dims) + (L,), device=dev,
                              dtype=torch.float64) for _ in range(3)]
            by = plan.algorithmic_bytes(8 * L, 8, engine.MODE_FRACB)
            for tag, tune in (('8 in flight', [10, 0, 0, 1, 0]),
                              ('16 in flight', [10, 0, 0, 1, 0, 16])):
                def run(i):
                    engine.remap_tensor(plan, m.dst_dims, xs[i % 3], [1],
                                        engine.MODE_FRACB, tune=tune,
                                        out=ys[i % 3])
                for i in range(5):
                    run(i)
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(30):
                    run(i)
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 30
                key = L
                same = None
                if key not in ref:
                    ref[key] = ys[0].clone()
                else:
                    same = bool(torch.equal(
                        torch.nan_to_num(ys[0], nan=-2.5),
                        torch.nan_to_num(ref[key], nan=-2.5)))
                print(json.dumps(dict(groups=shape, union=round(r, 3), L=L,
                                      variant=tag, ms=round(ms, 4),
                                      frac=round(by / (ms * 1e-3) / 8e12, 4),
                                      same=same)), flush=True)


if __name__ == '__main__':
    main()
