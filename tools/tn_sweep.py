#!/usr/bin/env python3
"""
(Time, nCells) and other short-run layouts: the lanes-across-rows kernel
(family 4, `spmm_rowcell`) against the permute-copy path and the
lane-per-(row, k) kernel, on config 3's mapping (mesh numbering).

    python tools/tn_sweep.py [--workload config3] [--locality mesh]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config3')
    ap.add_argument('--locality', default='mesh')
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--mode', default='fracb')
    ap.add_argument('--tiles', default='16x16,8x16,8x32,16x32')
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--only', default='',
                    help="'patch': only the LDS-staged variants, TT = 8, "
                         "shape (120, n_a) (for counter passes)")
    ap.add_argument('--tts', default='',
                    help='fields per lane to try (default 4,8,16; 8 with '
                         '--only patch)')
    ap.add_argument('--time', type=int, default=120, dest='n_time')
    ap.add_argument('--levels', default='',
                    help="'T:L,T:L': (T, n_a, L) shapes instead of the "
                         "default list")
    ap.add_argument('--no-cell', action='store_true',
                    help='skip the lanes-across-rows variants (long runs)')
    ap.add_argument('--sets', type=int, default=1,
                    help='distinct X buffers rotated over the launches '
                         '(>= 3 x 200 MB: Infinity-Cache-cold)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    m = synthetic.make_config(args.workload, device=dev,
                              locality=args.locality)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    plan.auto_schedule(m.dst_dims)
    mode = engine.MODE_MASKED if args.mode == 'masked' else engine.MODE_FRACB
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    shapes = [(12, m.n_a), (32, m.n_a), (120, m.n_a), (480, m.n_a),
              (60, m.n_a, 4), (40, m.n_a, 3), (120, m.n_a, 1)]
    if args.quick:
        shapes = [(12, m.n_a), (120, m.n_a), (60, m.n_a, 4)]
    if args.only == 'patch':
        shapes = [(args.n_time, m.n_a)]
        variants = []
    if args.levels:
        shapes = [(int(t), m.n_a, int(lv)) for t, lv in
                  (v.split(':') for v in args.levels.split(','))]
    variants = [('auto', None)] + [
        (f'rowcell tt={tt} unr={u}', [4, tt, u])
        for tt in (4, 8, 16) for u in (1, 2, 4)
        if (tt, u) not in ((4, 1), (16, 4))] + [('rowlane', [2]),
                                                ('rowgroup', [10])]
    if args.no_cell:
        variants = [('auto', None), ('rowgroup', [10])]
    for shape in shapes:
        x = torch.randn(shape, generator=g, device=dev, dtype=torch.float64)
        if args.mode == 'masked':
            x[:, torch.rand(m.n_a, generator=g, device=dev) < 0.2] = \
                float('nan')
        xs = [x] + [x.clone() for _ in range(args.sets - 1)]
        outs = [None] * args.sets
        turn = [0]
        K = x.numel() // m.n_a
        bytes_alg = plan.algorithmic_bytes(K, 8, mode)
        # the permute-copy path of round 2, for reference
        xt = x.movedim(1, 0).reshape(m.n_a, K)

        def permuted():
            X = xt.contiguous()
            y = engine.remap_tensor(plan, m.dst_dims, X, [0], mode,
                                    threshold=0.01)
            return y.reshape(m.n_b, K).t().contiguous()
        want = engine.remap_tensor(plan, m.dst_dims, x, [1], mode,
                                   threshold=0.01, tune=[1])
        print(f'{shape}: K = {K}, {bytes_alg / 1e6:.0f} MB algorithmic')
        if args.only != 'patch':
            t = timed(permuted, args.reps)
            print(f'   permute copies + (n_a, K) kernel   {t * 1e3:8.1f} us  '
                  f'{bytes_alg / t / 1e6 / 8000:.3f}')
        for name, tune in variants:
            if name == 'rowlane' and K > 32:
                continue

            def run():
                i = turn[0] = (turn[0] + 1) % args.sets
                outs[i] = engine.remap_tensor(plan, m.dst_dims, xs[i], [1],
                                              mode, threshold=0.01,
                                              tune=tune, out=outs[i])
                return outs[i]
            try:
                y = run()
            except engine.EngineError as exc:
                print(f'   {name:34s} {exc}')
                continue
            same = bool(((y == want) | (y.isnan() & want.isnan())).all())
            t = timed(run, args.reps)
            print(f'   {name:34s} {t * 1e3:8.1f} us  '
                  f'{bytes_alg / t / 1e6 / 8000:.3f}  '
                  f'{"bitwise" if same else "DIFFERS"}')
        # LDS-staged patches, lanes across rows (family 7)
        groups, order, tune0 = plan.groups, plan.row_order, plan.default_tune
        for tile in args.tiles.split(','):
            ty, tx = (int(v) for v in tile.split('x'))
            ratio = plan.build_patches(m.dst_dims, tile=(ty, tx),
                                       lds_budget=10 ** 9)
            plan.default_tune = None
            if plan.patches['umax'] * 1024 + plan.patches['emax'] * 12 \
                    < 150 * 1024:
                # the LDS patch kernel of family 5 (lanes across K)
                def run5():
                    i = turn[0] = (turn[0] + 1) % args.sets
                    outs[i] = engine.remap_tensor(
                        plan, m.dst_dims, xs[i], [1], mode, threshold=0.01,
                        tune=[5], out=outs[i])
                    return outs[i]
                y = run5()
                same = bool(((y == want) |
                             (y.isnan() & want.isnan())).all())
                t = timed(run5, args.reps)
                print(f'   patch (family 5) {tile} umax='
                      f'{plan.patches["umax"]:5d}          '
                      f'{t * 1e3:8.1f} us  '
                      f'{bytes_alg / t / 1e6 / 8000:.3f}  '
                      f'{"bitwise" if same else "DIFFERS"}')
            for tt in () if args.no_cell else (
                    tuple(int(v) for v in args.tts.split(',')) if args.tts
                    else (8,) if args.only == 'patch' else (4, 8, 16)):
                if plan.patches['umax'] * tt * 8 > 150 * 1024:
                    continue

                def run():
                    i = turn[0] = (turn[0] + 1) % args.sets
                    outs[i] = engine.remap_tensor(
                        plan, m.dst_dims, xs[i], [1], mode, threshold=0.01,
                        tune=[7, tt], out=outs[i])
                    return outs[i]
                y = run()
                same = bool(((y == want) |
                             (y.isnan() & want.isnan())).all())
                t = timed(run, args.reps)
                print(f'   patchcell {tile} tt={tt} umax='
                      f'{plan.patches["umax"]:5d} ratio={ratio:.2f}'
                      f'   {t * 1e3:8.1f} us  '
                      f'{bytes_alg / t / 1e6 / 8000:.3f}  '
                      f'{"bitwise" if same else "DIFFERS"}')
        plan.patches = None
        plan.row_order = order
        plan.groups = groups
        plan.default_tune = tune0


if __name__ == '__main__':
    main()
