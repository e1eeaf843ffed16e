#!/usr/bin/env python3
"""
The lock-step experiment: do re-touches of a source
row by NEIGHBOURING 2 x 2 groups hit L2 when they are made to coincide in
time by construction?

A workgroup owns one SUPERGROUP -- the 4 (4 x 4 destination cells) or 16
(8 x 8) groups of a supertile -- and its waves walk STEP-ALIGNED union lists:
step s covers the same range of source-row ids in every wave (at most 8
entries per wave and step; shorter steps are padded with entries whose member
mask is empty and whose load is a zero-byte buffer descriptor), with a
workgroup barrier per step.  The sums stay in ascending column order per row:
bitwise the same results.

    python tools/lockstep.py [--workload headline] [--super 4|8] [--pmc]

Prints the time per launch of the shipped schedule and of the aligned one;
under `rocprofv3 --pmc FETCH_SIZE` (tools/_lockstep_pmc.sh) the dispatches
come in the order printed.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402

INF = 2 ** 31 - 1


def align(groups, nw):
    """Step-align the union lists of every `nw` consecutive groups."""
    meta, col, mask = groups['meta'], groups['col'], groups['mask']
    dev = col.device
    n_g = groups['n']
    n_sg = (n_g + nw - 1) // nw
    start = meta[:-1, 0]
    lens = meta[1:, 0] - start
    pad_g = n_sg * nw - n_g
    if pad_g:
        start = torch.cat([start, start.new_zeros(pad_g)])
        lens = torch.cat([lens, lens.new_zeros(pad_g)])
    lmax = int(lens.max())
    ar = torch.arange(lmax + 9, device=dev)
    idx = start[:, None] + ar[None, :]
    valid = ar[None, :] < lens[:, None]
    idx = torch.where(valid, idx, torch.zeros_like(idx))
    L = torch.where(valid, col[idx].to(torch.int64),
                    torch.full_like(idx, INF))          # (n_sg * nw, lmax+9)
    M = torch.where(valid, mask[idx], torch.zeros_like(mask[idx]))
    L = L.reshape(n_sg, nw, -1)
    M = M.reshape(n_sg, nw, -1)
    lens = lens.reshape(n_sg, nw)
    p = torch.zeros_like(lens)
    a8 = torch.arange(8, device=dev)
    cols_it, masks_it, active_it = [], [], []
    while bool((p < lens).any()):
        ninth = torch.gather(L, 2, (p + 8).clamp(max=L.shape[2] - 1)[..., None])
        T = ninth.squeeze(-1).min(dim=1).values            # (n_sg,)
        win = (p[..., None] + a8).clamp(max=L.shape[2] - 1)
        wc = torch.gather(L, 2, win)
        wm = torch.gather(M, 2, win)
        take = wc < T[:, None, None]
        cols_it.append(torch.where(take, wc, torch.zeros_like(wc)))
        masks_it.append(torch.where(take, wm, torch.zeros_like(wm)))
        active_it.append((p < lens).any(dim=1))
        p = p + take.sum(-1)
    n_it = len(cols_it)
    C = torch.stack(cols_it, 2)          # (n_sg, nw, n_it, 8)
    Mk = torch.stack(masks_it, 2)
    act = torch.stack(active_it, 1)      # (n_sg, n_it)
    steps = act.sum(1)                   # (n_sg,)
    keep = act[:, None, :, None].expand_as(C)
    new_col = C[keep].to(torch.int32)
    new_mask = Mk[keep].to(torch.int32)
    per_group = (steps * 8)[:, None].expand(n_sg, nw).reshape(-1)[:n_g]
    new_meta = meta.clone()
    new_meta[1:, 0] = torch.cumsum(per_group, 0)
    new_meta[0, 0] = 0
    total = int(new_meta[-1, 0])
    # (groups beyond n_g do not exist: their -- empty -- lists are dropped)
    new_col = torch.cat([new_col[:total], new_col.new_zeros(32)])
    new_mask = torch.cat([new_mask[:total], new_mask.new_zeros(32)])
    out = dict(groups)
    out.update(meta=new_meta, col=new_col, mask=new_mask, union=total)
    real = int((new_mask[:total] != 0).sum())
    return out, dict(steps_mean=float(steps.double().mean()),
                     steps_max=n_it, slots=total, real=real,
                     fill=real / max(total, 1))


def timed(fn, reps):
    for _ in range(2):
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
    ap.add_argument('--workload', default='headline')
    ap.add_argument('--locality', default='mesh')
    ap.add_argument('--super', type=int, default=4, dest='st')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--mode', default='fracb')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    m = synthetic.make_config(args.workload, device=dev,
                              locality=args.locality)
    K = synthetic.CONFIGS[args.workload]['K']
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    mode = engine.MODE_MASKED if args.mode == 'masked' else engine.MODE_FRACB
    x = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
    if args.mode == 'masked':
        x[torch.rand(m.n_a, device=dev) < 0.25] = float('nan')
    bytes_alg = plan.algorithmic_bytes(K, 8, mode)
    out = [None]

    def run(tune):
        def f():
            out[0] = engine.remap_tensor(plan, m.dst_dims, x, [0], mode,
                                         threshold=0.01, tune=tune,
                                         out=out[0])
        return f

    def report(name, tune):
        ms = timed(run(tune), args.reps)
        print(f'{name:44s} {ms:8.4f} ms  {bytes_alg / ms / 1e6 / 8000:.3f} '
              f'(reads {(bytes_alg - m.n_b * K * 8) / ms / 1e6 / 8000:.3f})',
              flush=True)

    # 1. what ships: 2 x 2 groups, row-major
    plan.auto_schedule(m.dst_dims)
    run(None)()
    want = out[0].clone()
    report('shipped (2x2 groups, row-major, 4 waves/WG)', None)
    # 2. supertile walk, the same kernel (groups of a supertile consecutive)
    nw = (args.st // 2) ** 2
    plan.build_groups(m.dst_dims, super_tile=args.st, rows=4)
    report(f'{args.st}x{args.st} supertiles, plain walk', [10, 0, 1, 1])
    assert torch.equal(torch.nan_to_num(out[0], nan=1e300),
                       torch.nan_to_num(want, nan=1e300))
    # 3. step-aligned lists + barrier per step
    plain = plan.groups
    aligned, stats = align(plain, nw)
    print('aligned lists:', stats, 'plain union', plain['union'], flush=True)
    plan.groups = aligned
    report(f'{args.st}x{args.st} supergroups, lock-step ({nw} waves/WG)',
           [10, nw, 1, 1, 0, 108])
    assert torch.equal(torch.nan_to_num(out[0], nan=1e300),
                       torch.nan_to_num(want, nan=1e300)), 'DIFFERS'
    print('bitwise equal to the shipped schedule')


if __name__ == '__main__':
    main()
