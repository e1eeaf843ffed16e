#!/usr/bin/env python3
"""(Time, nCells, L) with short level runs (L = 6, 10, 12) on config 3's
map: the small LDS patches of RemapPlan.run_patches in other sizes and piece
widths.  GPU box only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    m = synthetic.make_config('config3', device=dev, locality='mesh')
    for L, T in ((10, 48), (6, 80), (12, 40)):
        xs = [torch.randn((T, m.n_a, L), device=dev, dtype=torch.float64)
              for _ in range(3)]
        ys = [torch.empty((T,) + tuple(m.dst_dims) + (L,), device=dev,
                          dtype=torch.float64) for _ in range(3)]
        ref = None
        for tile, row_bytes, budget in (((4, 8), 1024, 100), ((4, 8), 512, 100),
                                        ((8, 8), 512, 100), ((8, 16), 512, 100),
                                        ((8, 16), 512, 150), ((16, 16), 512, 150),
                                        ((8, 8), 1024, 150)):
            plan = engine.RemapPlan.from_triplets(
                m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, index_base=1,
                device=dev)
            plan.auto_schedule(m.dst_dims)

            def fits(rows, umax, emax, rb=row_bytes, bud=budget):
                return (umax + 1) * rb + emax * 12 + rows * 24 + 32 <= \
                    bud * 1024 or rows <= 4
            q = plan._make_patches(m.dst_dims, tile, fits, row_bytes)
            plan._runs = q
            plan._sched_version += 1
            by = plan.algorithmic_bytes(T * L, 8, engine.MODE_FRACB)

            def run(i):
                engine.remap_tensor(plan, m.dst_dims, xs[i % 3], [1],
                                    engine.MODE_FRACB, out=ys[i % 3])
            for i in range(4):
                run(i)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(20):
                run(i)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 20
            same = None
            if ref is None:
                ref = ys[0].clone()
            else:
                same = bool(torch.equal(torch.nan_to_num(ys[0], nan=-2.5),
                                        torch.nan_to_num(ref, nan=-2.5)))
            print(json.dumps(dict(L=L, T=T, tile=q['tile'], rows=q['rows'],
                                  umax=q['umax'], row_bytes=row_bytes,
                                  ms=round(ms, 4),
                                  frac=round(by / (ms * 1e-3) / 8e12, 4),
                                  same=same)), flush=True)


if __name__ == '__main__':
    main()
