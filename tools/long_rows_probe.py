#!/usr/bin/env python3
"""
Pole-capped global bilinear map (synthetic 'config1_esmf': 1 deg -> 0.5 deg
as ESMF makes it): GPU time of one apply replayed from a hipGraph -- the two
launches together, the short rows alone, the long rows alone -- for a few
field counts and layouts.  (GPU box only.)

    python tools/long_rows_probe.py [--workload config1_esmf]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def replay_us(launch, calls=40, reps=5):
    import torch
    launch()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(calls):
            launch()
    graph.replay()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * calls) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config1_esmf')
    ap.add_argument('--wave-fields', type=int, default=None,
                    help='engine.LONG_WAVE_FIELDS (family 9 up to that K)')
    ap.add_argument('--wave-rows', type=int, default=None,
                    help='engine.LONG_WAVE_ROWS (long rows per workgroup of '
                         'family 11; 0: family 7 beyond --wave-fields)')
    ap.add_argument('--tile', default=None,
                    help='engine.RemapPlan.CELL_TILE, e.g. 16x16')
    args = ap.parse_args()
    import torch

    from pyremap_amd import engine, synthetic
    if args.wave_fields is not None:
        engine.LONG_WAVE_FIELDS = args.wave_fields
    if args.tile:
        engine.RemapPlan.CELL_TILE = tuple(int(v)
                                           for v in args.tile.split('x'))
    if args.wave_rows is not None:
        engine.LONG_WAVE_ROWS = args.wave_rows
    dev = torch.device('cuda', 0)
    m = synthetic.make_config(args.workload, device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    print(json.dumps(plan.auto_schedule(m.dst_dims), default=str))
    short, long = plan._split
    cases = [('K1', (m.n_a,), [0], torch.float64),
             ('nk K12', (m.n_a, 12), [0], torch.float64),
             ('tn T12', (12, m.n_a), [1], torch.float64),
             ('nk K64', (m.n_a, 64), [0], torch.float64),
             ('tn T120 f32', (120, m.n_a), [1], torch.float32),
             ('nk K512', (m.n_a, 512), [0], torch.float64)]
    for tag, shape, axes, dtype in cases:
        x = torch.randn(shape, device=dev, dtype=torch.float64).to(dtype)
        y = engine.remap_tensor(plan, m.dst_dims, x, axes, engine.MODE_FRACB)

        def both():
            engine.remap_tensor(plan, m.dst_dims, x, axes, engine.MODE_FRACB,
                                out=y)
        saved = plan._split

        def only(which):
            def run():
                plan._split = (saved[0], None) if which == 0 else \
                    (None, saved[1])
                engine.remap_tensor(plan, m.dst_dims, x, axes,
                                    engine.MODE_FRACB, out=y)
                plan._split = saved
            return run
        row = dict(case=tag, both_us=round(replay_us(both), 2))
        for which, name in ((0, 'short_us'), (1, 'long_us')):
            try:
                row[name] = round(replay_us(only(which)), 2)
            except Exception as exc:   # noqa: BLE001
                plan._split = saved
                row[name] = str(exc)[:60]
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
