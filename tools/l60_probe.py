#!/usr/bin/env python3
"""(Time = 8, nCells, 60 levels) on config 3's map: tune variants of the
row-group kernel, cold (three field sets).  GPU box only."""
import json
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
    print(json.dumps(plan.auto_schedule(m.dst_dims), default=str))
    for L in (60, 64, 61):
        xs = [torch.randn((8, m.n_a, L), device=dev, dtype=torch.float64)
              for _ in range(3)]
        ys = [torch.empty((8,) + tuple(m.dst_dims) + (L,), device=dev,
                          dtype=torch.float64) for _ in range(3)]
        by = plan.algorithmic_bytes(8 * L, 8, engine.MODE_FRACB)
        ref = None
        variants = [('default', None), ('tiles 2', [10, 0, 2, 1, 0]),
                    ('tiles 1', [10, 0, 1, 1, 0]),
                    ('2 groups per wave', [10, 0, 0, 2, 0]),
                    ('16 in flight', [10, 0, 0, 1, 0, 16]),
                    ('rowscalar', [6]), ('rowscalar 2 tiles', [6, 0, 2])]
        for tag, tune in variants:
            def run(i):
                engine.remap_tensor(plan, m.dst_dims, xs[i % 3], [1],
                                    engine.MODE_FRACB, tune=tune,
                                    out=ys[i % 3])
            try:
                for i in range(5):
                    run(i)
            except engine.EngineError as exc:
                print(json.dumps(dict(L=L, variant=tag, error=str(exc)[:80])))
                continue
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(30):
                run(i)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 30
            same = None
            if ref is None:
                ref = ys[0].clone()
            else:
                same = bool(torch.equal(torch.nan_to_num(ys[0], nan=-2.5),
                                        torch.nan_to_num(ref, nan=-2.5)))
            print(json.dumps(dict(L=L, variant=tag, ms=round(ms, 4),
                                  frac=round(by / (ms * 1e-3) / 8e12, 4),
                                  same=same)), flush=True)


if __name__ == '__main__':
    main()
