#!/usr/bin/env python3
"""
(Time, nCells) fields in place on config 3's map: ms per launch of the
LDS-staged lanes-across-rows family (7) as the one-chunk kernel
(spmm_patchcell, tune[2] = 1) and as the workgroup-persistent one
(spmm_patchtime), Infinity-Cache-cold (three field sets rotated) and warm.

    python tools/tn_probe.py [--times 12,120] [--workload config3]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config3')
    ap.add_argument('--times', default='12,120')
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--dtype', default='f64')
    ap.add_argument('--tile', default='16x16',
                    help='tile of the destination grid per patch')
    ap.add_argument('--only', default='',
                    help='run only the variants whose tag contains this')
    args = ap.parse_args()
    import torch

    from pyremap_amd import engine, synthetic
    dev = torch.device('cuda', 0)
    m = synthetic.make_config(args.workload, device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    plan.auto_schedule(m.dst_dims)
    engine.RemapPlan.CELL_TILE = tuple(int(v) for v in args.tile.split('x'))
    q = plan.cell_patches()
    print(json.dumps(dict(tile=q['tile'], rows=q['rows'], umax=q['umax'],
                          patches=q['n'])))
    dt = torch.float32 if args.dtype == 'f32' else torch.float64
    for T in (int(t) for t in args.times.split(',')):
        xs = [torch.randn((T, m.n_a), device=dev, dtype=torch.float64).to(dt)
              for _ in range(3)]
        ys = [torch.empty((T,) + tuple(m.dst_dims), device=dev,
                          dtype=torch.float64) for _ in range(3)]
        by = plan.algorithmic_bytes(T, 4 if args.dtype == 'f32' else 8,
                                    engine.MODE_FRACB)
        ref = None
        for tag, tune in (('one chunk per workgroup, TT 8', [8, 1]),
                          ('persistent, TT 8', [8, 0]),
                          ('persistent, TT 8, 1 run', [8, 0, 1]),
                          ('persistent, TT 4', [4, 0]),
                          ('persistent, TT 4, 1 run', [4, 0, 1]),
                          ('persistent, TT 4, 4 runs', [4, 0, 4]),
                          ('persistent, TT 2, 1 run', [2, 0, 1]),
                          ('persistent, TT 2, 2 runs', [2, 0, 2]),
                          ('default', None)):
            if args.only and args.only not in tag:
                continue
            engine._CELL_TUNE = tune
            for sets, label in ((3, 'cold'), (1, 'warm')):
                def run(i):
                    engine.remap_tensor(plan, m.dst_dims, xs[i % sets], [1],
                                        engine.MODE_FRACB, out=ys[i % sets])
                for i in range(5):
                    run(i)
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(args.reps):
                    run(i)
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / args.reps
                same = None
                if ref is None:
                    ref = ys[0].clone()
                else:
                    same = bool(torch.equal(
                        torch.nan_to_num(ys[0], nan=-2.5),
                        torch.nan_to_num(ref, nan=-2.5)))
                print(json.dumps(dict(
                    T=T, kernel=tag, cache=label, ms=round(ms, 4),
                    frac=round(by / (ms * 1e-3) / 8e12, 4),
                    bitwise_equal=same)), flush=True)


if __name__ == '__main__':
    main()
