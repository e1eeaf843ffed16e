#!/usr/bin/env python3
"""
Interleaved A/B sweep of launch shapes for one workload (GPU box only).

    python tools/sweep.py --workload config3 --tunes "0;1,2,1,4,2;1,2,2,4,2"

Variants are timed round-robin in ONE process (several rounds, HIP events on
the launch stream), the table reports median / min per variant -- the
methodology of cdna_hip_programming.md section 5.4 rule 24.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config3')
    ap.add_argument('--fields', type=int, default=None)
    ap.add_argument('--mode', default='fracb')
    ap.add_argument('--locality', default='raster')
    ap.add_argument('--mask', default='cells',
                    choices=['cells', 'levels', 'coast', 'few'],
                    help='masked mode: which values are NaN')
    ap.add_argument('--tunes', default='0')
    ap.add_argument('--flags', default='0')
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--sets', type=int, default=2)
    ap.add_argument('--dtype', default='f64')
    ap.add_argument('--lds-kb', type=int, default=80)
    ap.add_argument('--x-pad', type=int, default=0,
                    help='extra elements between the rows of X')
    ap.add_argument('--y-pad', type=int, default=0,
                    help='extra elements between the rows of Y')
    ap.add_argument('--pmc', type=int, default=0,
                    help='counter mode: launch each variant N times in '
                         'order, no timing (run under rocprofv3 --pmc)')
    ap.add_argument('--shard', default=None,
                    help="R/N: time the rows rank R of N would own")
    ap.add_argument('--levels', type=int, default=0,
                    help='address the (n_a, K) buffers as K / L batches of L '
                         'levels: n_batch = K / L, k_inner = L, batch stride '
                         'L (what REMAP_FLAG_BATCH_MASKS wants to see)')
    ap.add_argument('--tnl', type=int, default=0,
                    help='read the buffers as (T, nCells, K / T) fields -- '
                         "MPAS's layout: n_batch = T, k_inner = K / T, batch "
                         'stride n_a * K / T (the mask options then apply to '
                         'that layout)')
    ap.add_argument('--pairs', default=None,
                    help="';'-separated order|tune[|flags]: the variants, "
                         'instead of the product of --orders x --flags x '
                         '--tunes (the orders still come from --orders)')
    ap.add_argument('--orders', default='none',
                    help="';'-separated: none | morton | tile:TYxTX")
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = synthetic.CONFIGS[args.workload]
    K = args.fields or cfg['K']
    m = synthetic.make_config(args.workload, device=dev,
                              locality=args.locality)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    if args.shard:
        r, n = (int(v) for v in args.shard.split('/'))
        plan = plan.shard(r, n)
        print(f'shard {r}/{n}: rows [{plan.row_offset}, '
              f'{plan.row_offset + plan.n_b}) nnz {plan.nnz}')
    mode = {'fracb': engine.MODE_FRACB, 'masked': engine.MODE_MASKED,
            'raw': engine.MODE_RAW}[args.mode]
    dt = torch.float64 if args.dtype == 'f64' else torch.float32
    xs = [torch.randn((m.n_a, K + args.x_pad), device=dev, dtype=dt)
          for _ in range(args.sets)]
    if args.mode == 'masked' and args.tnl:
        for x in xs:
            x3 = x.view(args.tnl, m.n_a, K // args.tnl)
            x3[:, torch.rand(m.n_a, device=dev) < 0.25, :] = float('nan')
            if args.mask == 'levels':
                Lt = K // args.tnl
                depth = torch.randint(8, Lt + 1, (m.n_a, 1), device=dev)
                lev = torch.arange(Lt, device=dev)[None]
                x3.masked_fill_((lev >= depth)[None], float('nan'))
    elif args.mode == 'masked':
        for x in xs:
            if args.mask == 'cells':      # whole cells missing (land)
                x[torch.rand(m.n_a, device=dev) < 0.25, :] = float('nan')
            elif args.mask == 'levels':
                # (Time, level) columns, 64 levels: a cell is missing below
                # its own depth (bathymetry) -- validity differs by lane
                depth = torch.randint(8, 65, (m.n_a, 1), device=dev)
                lev = (torch.arange(K + args.x_pad, device=dev) % 64)[None]
                x[lev >= depth] = float('nan')
            elif args.mask == 'coast':
                # land = one corner of the source numbering (a contiguous
                # fifth of the cells): most groups see no NaN at all
                x[: m.n_a // 5, :] = float('nan')
            elif args.mask == 'few':
                x[torch.rand(m.n_a, device=dev) < 0.001, :] = float('nan')
    ys = [torch.empty((plan.n_b, K + args.y_pad), device=dev,
                      dtype=torch.float64)
          for _ in range(args.sets)]
    variants = []
    orders = {}
    patch_sets = {}
    group_sets = {}
    for o in args.orders.split(';'):
        if o.startswith('group'):
            # group:<supertile>[:<rows per group>[:share<W>]] -- share<W>:
            # + the shared union lists of W groups (spmm_groupshare.h)
            parts = o.split(':')
            st = int(parts[1]) if len(parts) > 1 else 32
            share = int(parts[3][5:]) if len(parts) > 3 else 0
            ratio = plan.build_groups(
                None if o.startswith('group1d') else m.dst_dims,
                super_tile=st, rows=int(parts[2]) if len(parts) > 2 else 8,
                share=share)
            print(f'{o}: union/nnz = {ratio}' + (
                f', shared by {share} groups: '
                f'{plan.groups["share"]["ratio"]}' if share else ''))
            orders[o] = plan.row_order
            group_sets[o] = plan.groups
            continue
        if o.startswith('patch'):
            ty, tx = o.split(':')[1].split('x')
            ratio = plan.build_patches(
                m.dst_dims, tile=(int(ty), int(tx)),
                lds_budget=args.lds_kb * 1024,
                row_bytes=512 if o.startswith('patch512') else 1024)
            print(f'{o}: distinct/nnz = {ratio}, '
                  f'tile {plan.patches["tile"]}, umax '
                  f'{plan.patches["umax"]}, patches {plan.patches["n"]}')
            orders[o] = plan.row_order
            patch_sets[o] = plan.patches
            continue
        if o == 'none':
            orders[o] = None
        elif o == 'morton':
            plan.set_grid_schedule(m.dst_dims, 'morton')
            orders[o] = plan.row_order
        else:
            ty, tx = o.split(':')[1].split('x')
            plan.set_grid_schedule(m.dst_dims, 'tile', (int(ty), int(tx)))
            orders[o] = plan.row_order
    for o in orders:
        for fl in args.flags.split(';'):
            for t in args.tunes.split(';'):
                tune = [int(v) for v in t.split(',')] if t != '0' else None
                variants.append((int(fl), tune, o))
    if args.pairs:
        variants = []
        for pr in args.pairs.split(';'):
            parts = pr.split('|')
            t = parts[1]
            variants.append((int(parts[2]) if len(parts) > 2 else 0,
                             [int(v) for v in t.split(',')] if t != '0'
                             else None, parts[0]))
    bytes_alg = plan.algorithmic_bytes(K, xs[0].element_size(), mode)
    print(f'{args.workload}: n_a={m.n_a} n_b={m.n_b} nnz={plan.nnz} K={K} '
          f'bytes_alg={bytes_alg / 1e9:.3f} GB mode={args.mode}')

    def launch(v, i):
        fl, tune, o = v
        plan.row_order = orders[o]
        plan.patches = patch_sets.get(o)
        plan.groups = group_sets.get(o)
        s = i % args.sets
        if args.tnl:
            Lt = K // args.tnl
            engine.apply_strided(plan, xs[s], ys[s], n_batch=args.tnl,
                                 k_inner=Lt, x_row_stride=Lt,
                                 x_batch_stride=m.n_a * Lt, y_row_stride=Lt,
                                 y_batch_stride=plan.n_b * Lt, mode=mode,
                                 threshold=0.01, flags=fl, tune=tune)
            return
        L = args.levels or K
        engine.apply_strided(plan, xs[s], ys[s], n_batch=K // L, k_inner=L,
                             x_row_stride=K + args.x_pad,
                             x_batch_stride=L if args.levels else 0,
                             y_row_stride=K + args.y_pad,
                             y_batch_stride=L if args.levels else 0,
                             mode=mode,
                             threshold=0.01, flags=fl, tune=tune)

    if args.pmc:
        for vi, v in enumerate(variants):
            print(f'PMCVARIANT {vi} {v[2]} flags={v[0]} tune={v[1]}')
            for i in range(args.pmc):
                launch(v, i)
            torch.cuda.synchronize()
        return
    times = {i: [] for i in range(len(variants))}
    ref = None
    for vi, v in enumerate(variants):
        try:
            launch(v, 0)
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            print(f'variant {v}: {exc}')
            times[vi] = None
            continue
        if not (v[0] & 1) and not (v[1] and len(v[1]) > 6 and v[1][6]):
            if ref is None:
                ref = ys[0].clone()
            else:
                # bit for bit (no temporaries: config 5's Y is 53 GB)
                same = torch.equal(ys[0].view(torch.int64),
                                   ref.view(torch.int64))
                assert same, f'variant {v} changes the result'
    for rnd in range(args.rounds):
        for vi, v in enumerate(variants):
            if times[vi] is None:
                continue
            launch(v, 1)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(args.reps):
                launch(v, i)
            b.record()
            torch.cuda.synchronize()
            times[vi].append(a.elapsed_time(b) / args.reps)
    print(f'{"order":<12} {"flags":>5} {"tune":<22} {"med ms":>8} {"min ms":>8} '
          f'{"GB/s(med)":>10} {"%8TB/s":>7}')
    for vi, v in enumerate(variants):
        if not times[vi]:
            continue
        ts = sorted(times[vi])
        med, mn = ts[len(ts) // 2], ts[0]
        gbps = bytes_alg / (med * 1e-3) / 1e9
        print(f'{v[2]:<12} {v[0]:>5} {str(v[1]):<22} {med:8.4f} {mn:8.4f} '
              f'{gbps:10.1f} {gbps / 80:7.1f}')


if __name__ == '__main__':
    main()
