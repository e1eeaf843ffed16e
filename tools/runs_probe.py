#!/usr/bin/env python3
"""(Time, nCells, L) with short level runs (4 <= L < 16) on config 3's map:
the default route (L <= 6: the batch-at-a-time lanes-across-rows kernel
spmm_patchtime<..., RUNS> on 256-row patches; 7 <= L < 16: the small LDS
patches of family 5), the batch-at-a-time kernel forced at every L
(RUN_CELLS_MAX raised), the small LDS patches forced (4 x 8 tiles, 1 KiB
pieces) and the row groups.  GPU box only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def timed(run, n=20):
    for i in range(4):
        run(i)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        run(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device('cuda', 0)
    m = synthetic.make_config('config3', device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    plan.auto_schedule(m.dst_dims)
    old = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                         m.n_b, index_base=1, device=dev)
    q = old._make_patches(
        m.dst_dims, (4, 8), lambda rows, umax, emax:
        (umax + 1) * 1024 + emax * 12 + rows * 24 + 32 <= 100 * 1024 or
        rows <= 4, 1024)
    old.row_order = q['order']
    old.patches = q
    forced = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                            m.n_a, m.n_b, index_base=1,
                                            device=dev)
    forced.auto_schedule(m.dst_dims)
    forced.RUN_CELLS_MAX = 15
    for L, T in ((4, 120), (5, 96), (6, 80), (8, 60), (10, 48), (12, 40),
                 (15, 32)):
        xs = [torch.randn((T, m.n_a, L), device=dev, dtype=torch.float64)
              for _ in range(3)]
        ys = [torch.empty((T,) + tuple(m.dst_dims) + (L,), device=dev,
                          dtype=torch.float64) for _ in range(3)]
        by = plan.algorithmic_bytes(T * L, 8, engine.MODE_FRACB)
        row = dict(L=L, T=T)
        ref = None
        for tag, p, tune in (('default', plan, None),
                             ('batch_single_loads', forced, 'single'),
                             ('batch_at_a_time', forced, None),
                             ('lds_patches_4x8', old, [5]),
                             ('row_groups', plan, [10, 0, 0, 1, 0]),
                             ('rowlane', plan, [2]),
                             ('rowcell', plan, [4]),
                             ('rowscalar', plan, [6])):
            # ('single': the batch-at-a-time kernel with one element per
            # load -- round 4's form -- instead of aligned pairs)
            engine._RUNS_TUNE = [7, 4, 2, 0, 0, 1] if tune == 'single' \
                else None
            if tune == 'single':
                tune = None

            def run(i):
                engine.remap_tensor(p, m.dst_dims, xs[i % 3], [1],
                                    engine.MODE_FRACB, tune=tune,
                                    out=ys[i % 3])
            try:
                ms = timed(run)
            except engine.EngineError as exc:
                row[tag + '_error'] = str(exc)[:60]
                continue
            if ref is None:
                ref = ys[0].clone()
            else:
                row[tag + '_same'] = bool(torch.equal(
                    torch.nan_to_num(ys[0], nan=-2.5),
                    torch.nan_to_num(ref, nan=-2.5)))
            row[tag + '_ms'] = round(ms, 4)
            row[tag + '_frac'] = round(by / (ms * 1e-3) / 8e12, 4)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
