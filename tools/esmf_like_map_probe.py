#!/usr/bin/env python3
"""
A global lat-lon -> finer lat-lon bilinear map as ESMF makes it
(`pyremap_amd.weights.bilinear_3d`) carries two rows of destination cells at
either pole whose entries are the WHOLE adjacent source row (the pole cap):
360 + 2 entries per row among rows of 4.  Which schedule does the mapping
get, is the result the oracle's, and what does a launch cost next to the
synthetic config-1 map of the same size without caps?  (GPU box only.)

    python tools/esmf_like_map_probe.py [--src 1.0 --dst 0.5 --fields 128]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pyremap_amd import engine, get_lat_lon_descriptor  # noqa: E402
from pyremap_amd.weights import build_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--src', type=float, default=1.0)
    ap.add_argument('--dst', type=float, default=0.5)
    ap.add_argument('--fields', type=int, default=128)
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--long-tt', type=int, default=0)
    args = ap.parse_args()
    engine._LONG_TT = args.long_tt or None
    dev = torch.device('cuda', 0)
    src = get_lat_lon_descriptor(args.src, args.src)
    dst = get_lat_lon_descriptor(args.dst, args.dst)
    t0 = time.perf_counter()
    m = build_weights(src, dst, 'bilinear')
    print(f'weights: {time.perf_counter() - t0:.1f} s, n_a {m.n_a}, n_b '
          f'{m.n_b}, n_s {len(m.S)}')
    rows = np.bincount(m.row - 1, minlength=m.n_b)
    print(f'entries per row: median {int(np.median(rows))}, max '
          f'{rows.max()}, rows > 8: {(rows > 8).sum()} holding '
          f'{rows[rows > 8].sum() / rows.sum():.1%} of the entries')
    dims = tuple(int(d) for d in m.dst_grid_dims[::-1])
    K = args.fields
    for label, keep in (('with the pole caps', np.ones(len(m.S), bool)),
                        ('cap rows removed', rows[m.row - 1] <= 8)):
        plan = engine.RemapPlan.from_triplets(
            m.row[keep], m.col[keep], m.S[keep], m.frac_b, m.n_a, m.n_b,
            device=dev)
        choice = plan.auto_schedule(dims)
        xs = [torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
              for _ in range(3)]
        ys = [torch.empty(dims + (K,), device=dev, dtype=torch.float64)
              for _ in range(3)]
        for layout, axes in (('(n_a, K)', [0]),):
            for i in range(6):
                engine.remap_tensor(plan, dims, xs[i % 3], axes,
                                    engine.MODE_FRACB, out=ys[i % 3])
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(60):
                engine.remap_tensor(plan, dims, xs[i % 3], axes,
                                    engine.MODE_FRACB, out=ys[i % 3])
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 60
            by = plan.algorithmic_bytes(K, 8, engine.MODE_FRACB)
            if 'long_rows' in choice:
                label += f' (+{choice["long_rows"]} long rows apart)'
            print(f'{label:20s} schedule {choice["family"]:10s} '
                  f'{ms * 1e3:8.1f} us  {by / ms / 1e6 / 8000:.3f} of 8 TB/s')
        if args.check and label.startswith('with'):
            from oracle import oracle
            rowptr, col, val = plan.to_host_csr()
            csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
            want, mask = oracle.remap_flat(csr, m.frac_b, xs[0].cpu().numpy(),
                                           False, 0.0)
            want[mask] = np.nan
            got = ys[0].reshape(m.n_b, K).cpu().numpy()
            print('   == oracle bitwise:',
                  bool(np.array_equal(got, want, equal_nan=True)))


if __name__ == '__main__':
    main()
