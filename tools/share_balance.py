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
    pop8 = np.array([bin(i).count('1') for i in range(256)], dtype=np.int64)
    tot_mean = tot_max = 0.0
    n_steps = 0
    own_hist = np.zeros(9, dtype=np.int64)
    lens = []
    for s in pick:
        lo, hi = meta[s], meta[s + 1]
        mk = mask[lo:hi]
        n = hi - lo
        lens.append(n)
        pad = (-n) % 8
        if pad:
            mk = np.concatenate([mk, np.zeros(pad, dtype=np.uint32)])
        by = np.stack([(mk >> (8 * w)) & 0xff for w in range(4)], axis=1)
        prod = pop8[by].reshape(-1, 8, 4).sum(axis=1)          # (steps, 4)
        own = (by != 0).reshape(-1, 8, 4).sum(axis=1)
        cost = args.c_step + args.c_own * own + args.c_prod * prod
        tot_mean += cost.mean(axis=1).sum()
        tot_max += cost.max(axis=1).sum()
        n_steps += cost.shape[0]
        own_hist += np.bincount(own.ravel(), minlength=9)[:9]
    lens = np.asarray(lens)
    print(f'{args.workload}: {len(pick)} of {n_super} supergroups, '
          f'{n_steps} steps; union entries per supergroup: mean '
          f'{lens.mean():.1f}, median {np.median(lens):.0f}, max {lens.max()}')
    print(f'owned entries per wave and step (0..8): '
          f'{(own_hist / own_hist.sum()).round(3).tolist()}')
    print(f'cost per step: mean over waves {tot_mean / n_steps:.1f}, mean of '
          f'the slowest wave {tot_max / n_steps:.1f} '
          f'(x {tot_max / tot_mean:.3f})')


if __name__ == '__main__':
    main()
