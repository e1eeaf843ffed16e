#!/usr/bin/env python3
"""
Randomised stress of the host-array pipeline (GPU box only): many calls with
random shapes / layouts / modes / chunk sizes, several in flight at once,
every result against the CPU oracle.  Looks for races between the helper
upload thread, the three streams and the pinned-buffer bookkeeping.

    python tools/stress_host_path.py [--calls 300] [--seed 0]
"""
import argparse
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from pyremap_amd import engine, host_path, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--calls', type=int, default=300)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    dev = torch.device('cuda', 0)
    maps = []
    for seed, (n_a, dims, lo, hi) in enumerate((
            (3000, (40, 50), 1, 6), (5000, (30, 70), 2, 12),
            (800, (25, 31), 1, 4))):
        m = synthetic.conservative_map(n_a, dims, lo, hi, seed=seed,
                                       signed=seed == 1)
        mm = m.numpy()
        plan = engine.RemapPlan.from_triplets(
            mm['row'], mm['col'], mm['S'], mm['frac_b'], m.n_a, m.n_b,
            device=dev)
        plan.auto_schedule(m.dst_dims)
        csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'],
                                m.n_b, m.n_a)
        maps.append((m, mm, plan, csr))
    base = host_path.pinned_bytes_alive()
    bad = 0
    inflight = []
    for call in range(args.calls):
        m, mm, plan, csr = maps[rng.integers(len(maps))]
        layout = rng.integers(4)
        K = int(rng.choice([1, 5, 40, 64, 130, 257]))
        if layout == 0:
            shape, axes = (m.n_a, K), [0]
        elif layout == 1:
            shape, axes = (int(rng.integers(2, 7)), m.n_a, max(K, 8)), [1]
        elif layout == 2:
            shape, axes = (m.n_a,), [0]
        else:
            shape, axes = (K, m.n_a), [1]        # permute path for K > 32
        dtype = np.float32 if rng.random() < 0.3 else np.float64
        x = rng.standard_normal(shape).astype(dtype)
        mode = ['fracb', 'masked', 'auto'][rng.integers(3)]
        if mode != 'fracb' and rng.random() < 0.8:
            idx = [slice(None)] * len(shape)
            idx[axes[0]] = rng.random(m.n_a) < 0.2
            x[tuple(idx)] = np.nan
        thr = None if mode == 'fracb' else float(rng.choice([0.0, 0.1, 0.6]))
        host_path.CHUNK_BYTES = int(rng.choice([16, 64, 256, 4096])) * 1024
        want_mask = mode != 'auto' and rng.random() < 0.5
        pend = host_path.remap_host_array(plan, m.dst_dims, x, axes,
                                          mode=mode, threshold=thr,
                                          want_mask=want_mask)
        masked = mode == 'masked' or (mode == 'auto' and
                                      bool(np.isnan(x).any()))
        inflight.append((pend, x, axes, masked, thr, want_mask, m, mm, csr,
                         mode, shape))
        if len(inflight) >= int(rng.integers(1, 5)):
            for (pend, x, axes, masked, thr, want_mask, m, mm, csr, mode,
                 shape) in inflight:
                arg = np.ma.masked_array(x, np.isnan(x)) if masked else x
                ref = oracle.remap_numpy_array(
                    csr, mm['frac_b'], m.dst_dims, arg, axes,
                    thr if masked else None)
                got = pend.result()
                data = got[0] if want_mask else got
                exp = np.ma.filled(ref, np.nan)
                same = data.shape == exp.shape and np.array_equal(
                    np.isnan(data), np.isnan(exp)) and np.array_equal(
                    data[~np.isnan(exp)].view(np.int64),
                    exp[~np.isnan(exp)].view(np.int64))
                if want_mask:
                    same = same and np.array_equal(
                        got[1], np.ma.getmaskarray(ref))
                if not same:
                    bad += 1
                    print('MISMATCH', mode, shape, x.dtype, axes, thr)
            inflight = []
            gc.collect()
    del pend, got, data
    gc.collect()
    print(f'{args.calls} calls, {bad} mismatches, pinned bytes alive '
          f'{host_path.pinned_bytes_alive() - base}')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
