#!/usr/bin/env python3
"""Where do 60-level runs lose?  (8, nCells, 60) on config 3's map with the
source and / or the destination rows PITCHED to 64 elements (whole cache
lines): apply_strided with explicit strides.  GPU box only."""
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
    plan.auto_schedule(m.dst_dims)
    T, L = 8, 60
    by = plan.algorithmic_bytes(T * L, 8, engine.MODE_FRACB)
    for xp in (60, 64):
        for yp in (60, 64):
            xs = [torch.randn((T, m.n_a, xp), device=dev, dtype=torch.float64)
                  for _ in range(3)]
            ys = [torch.empty((T, m.n_b, yp), device=dev, dtype=torch.float64)
                  for _ in range(3)]

            def run(i):
                engine.apply_strided(
                    plan, xs[i % 3], ys[i % 3], n_batch=T, k_inner=L,
                    x_row_stride=xp, x_batch_stride=m.n_a * xp,
                    y_row_stride=yp, y_batch_stride=m.n_b * yp,
                    mode=engine.MODE_FRACB)
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
            print(json.dumps(dict(x_pitch=xp, y_pitch=yp, ms=round(ms, 4),
                                  frac=round(by / (ms * 1e-3) / 8e12, 4))),
                  flush=True)


if __name__ == '__main__':
    main()
