#!/usr/bin/env python3
"""
How evenly does a step of the shared form (csrc/spmm_groupshare.h) load the
four waves of a workgroup?  Per step of 8 union entries every wave adds the
entries ITS 8 rows own (`owned`), one product per (entry, member row)
(`products`), then all four meet at a barrier: the step lasts as long as its
slowest wave.  Prints, over a sample of supergroups of a workload's schedule,
the mean per-wave cost of a step and the mean of the per-step maximum, for a
cost of `c_own` instructions per owned entry and `c_prod` per product.

    python tools/share_balance.py [--workload config5] [--sample 20000]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config5')
    ap.add_argument('--sample', type=int, default=20000)
    ap.add_argument('--c-own', type=float, default=8.0)
    ap.add_argument('--c-prod', type=float, default=13.0)
    ap.add_argument('--c-step', type=float, default=60.0)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    m = synthetic.make_config(args.workload, device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    plan.build_groups(m.dst_dims, super_tile=32, rows=8, share=4)
    sh = plan.groups['share']
    meta = sh['meta'][:, 0].cpu().numpy()
    n_super = meta.shape[0] - 1
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(n_super, min(args.sample, n_super),
                              replace=False))
    mask = sh['mask'].cpu().numpy().view(np.uint32)
    # what-if: the 32 slots of a supergroup dealt to the waves differently
    # (slot q of the tile walk -> wave perm[q] // 8, member perm[q] % 8)
    q = np.arange(32)
    # slot q of the walk -> its place (py, px) in the 4 x 8 tile (order_keys:
    # wave = 2 x 4 patch (wy, wx), member = row-major inside the patch)
    py = (q // 8 // 2) * 2 + (q % 8) // 4
    px = (q // 8 % 2) * 4 + q % 4
    perms = {
        'as built (a 2 x 4 patch per wave)': q,
        'columns: wave px % 4': (px % 4) * 8 + py * 2 + px // 4,
        'checkerboard: wave (py % 2, px % 2)':
            ((py % 2) * 2 + px % 2) * 8 + (py // 2) * 4 + px // 2,
        'diagonal: wave (px + py) % 4': ((px + py) % 4) * 8 + py * 2 + px // 4,
        'diagonal 2: wave (px + 2 py) % 4':
            ((px + 2 * py) % 4) * 8 + py * 2 + px // 4,
        'rows: wave py': py * 8 + px,
        'pairs of columns: wave (px // 2) % 4':
            ((px // 2) % 4) * 8 + py * 2 + px % 2,
    }
    for name, perm in perms.items():
        assert sorted(perm.tolist()) == list(range(32)), name
    cat = [mask[meta[s]:meta[s + 1]] for s in pick]
    lens = np.asarray([len(c) for c in cat])
    padded = []
    for c in cat:
        pad = (-len(c)) % 8
        padded.append(np.concatenate([c, np.zeros(pad, dtype=np.uint32)])
                      if pad else c)
    allm = np.concatenate(padded)
    bits = ((allm[:, None] >> q[None, :]) & 1).astype(np.int64)  # (E, 32)
    print(f'{args.workload}: {len(pick)} of {n_super} supergroups, '
          f'{len(allm) // 8} steps; union entries per supergroup: mean '
          f'{lens.mean():.1f}, median {np.median(lens):.0f}, max {lens.max()}')
    for name, perm in perms.items():
        nb = np.zeros_like(bits)
        nb[:, perm] = bits
        per_wave = nb.reshape(-1, 4, 8)                    # (E, wave, member)
        prod = per_wave.sum(axis=2).reshape(-1, 8, 4).sum(axis=1)
        own = (per_wave.sum(axis=2) != 0).reshape(-1, 8, 4).sum(axis=1)
        cost = args.c_step + args.c_own * own + args.c_prod * prod
        hist = np.bincount(own.ravel(), minlength=9)[:9]
        print(f'{name}: owned per wave and step {own.mean():.2f} '
              f'(0..8: {(hist / hist.sum()).round(2).tolist()}); cost per '
              f'step: mean over waves {cost.mean():.1f}, mean of the slowest '
              f'{cost.max(axis=1).mean():.1f} '
              f'(x {cost.max(axis=1).mean() / cost.mean():.3f})')


if __name__ == '__main__':
    main()
