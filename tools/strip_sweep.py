#!/usr/bin/env python3
"""
Config 5 (or any entry-rich synthetic config) through kernel family 8 -- the
LDS ring sliding along strips of the destination grid -- for a list of strip
shapes, beside the plan's automatic schedule in the same process.

    python tools/strip_sweep.py [--workload config5] [--fields K]
        [--shapes R,W,SEG,DEPTH ...] [--masked] [--check]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(launch, reps):
    import torch
    for _ in range(2):
        launch()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        launch()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config5')
    ap.add_argument('--fields', type=int, default=None)
    ap.add_argument('--shapes', nargs='*',
                    default=['8,1,4,2,8', '8,1,4,3,8', '8,2,4,2,8', '8,2,4,3,8',
                             '14,1,4,2,14', '16,1,4,2,8', '16,1,4,2,14',
                             '6,1,4,2,6', '12,1,4,2,12'],
                    help='strip_rows,step_cols,segments,depth,waves[,gap]')
    ap.add_argument('--masked', action='store_true')
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--flags', type=int, default=0)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import torch

    from pyremap_amd import engine, strips, synthetic
    dev = torch.device('cuda', 0)
    cfg = synthetic.CONFIGS[args.workload]
    K = args.fields or cfg['K']
    m = synthetic.make_config(args.workload, device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    sched = plan.auto_schedule(m.dst_dims)
    mode = engine.MODE_MASKED if args.masked else engine.MODE_FRACB
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x = torch.randn((m.n_a, K), generator=g, device=dev, dtype=torch.float64)
    if args.masked:
        dead = torch.rand(m.n_a, generator=g, device=dev) < 0.25
        x[dead] = float('nan')
    y = torch.empty((m.n_b, K), device=dev, dtype=torch.float64)
    bytes_alg = plan.algorithmic_bytes(K, 8, mode)
    rows = []

    def report(tag, ms, **kw):
        row = dict(tag=tag, ms=round(ms, 4),
                   frac=round(bytes_alg / (ms * 1e-3) / 8e12, 4), **kw)
        rows.append(row)
        print(json.dumps(row), flush=True)

    def auto():
        engine.remap_tensor(plan, None, x, [0], mode, threshold=0.01,
                            flags=args.flags, out=y)
    report('auto: ' + sched['family'], timed(auto, args.reps))
    ref = y.clone() if args.check else None
    for shape in args.shapes:
        R, W, S, D, NW, *rest = (int(v) for v in shape.split(','))
        G = rest[0] if rest else 1
        t0 = time.perf_counter()
        try:
            st = plan.build_strips(m.dst_dims, strip_rows=R, step_cols=W,
                                   segments=S, depth=D, gap=G,
                                   waves=NW)
        except strips.StripsUnfit as exc:
            print(json.dumps(dict(tag=shape, error=str(exc))), flush=True)
            continue
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0

        def strip():
            engine.remap_tensor(plan, None, x, [0], mode, threshold=0.01,
                                flags=args.flags, tune=[8], out=y)
        ms = timed(strip, args.reps)
        same = None
        if args.check:
            same = bool(torch.equal(torch.nan_to_num(y, nan=-1.5),
                                    torch.nan_to_num(ref, nan=-1.5)))
        report(f'strips R={R} W={W} seg={S} depth={D} waves={NW} gap={G}', ms,
               ring_slots=st['ring_slots'], lds_bytes=st['lds_bytes'],
               arrivals_per_source_row=round(st['arrivals'] / m.n_a, 3),
               build_s=round(build_s, 2), bitwise_equal_to_auto=same)
        plan.strips = None
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
