#!/usr/bin/env python3
"""Where the one-off cost of a Remapper goes: file read, COO -> CSR,
schedule (GPU box)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402
from pyremap_amd.io import mapfile  # noqa: E402


def t(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


for name in sys.argv[1:] or ['config3', 'headline']:
    m = synthetic.make_config(name, device='cuda:0')
    host = m.numpy()
    with tempfile.TemporaryDirectory() as tmp:
        for fmt in ('NETCDF3_64BIT', 'NETCDF4'):
            path = os.path.join(tmp, f'map_{fmt}.nc')
            dt_w, _ = t(lambda: mapfile.write_mapping(
                path, m.n_a, m.n_b, host['src_grid_dims'],
                host['dst_grid_dims'], host['row'], host['col'], host['S'],
                host['frac_b'], format=fmt))
            dt_r, mf = t(lambda: mapfile.read_mapping(path))
            print(f'{name} {fmt}: write {dt_w:.3f} s, read {dt_r:.3f} s '
                  f'({os.path.getsize(path) / 1e6:.0f} MB)')
    dt_c, plan = t(lambda: engine.RemapPlan.from_triplets(
        mf.row, mf.col, mf.S, mf.frac_b, mf.n_a, mf.n_b, device='cuda:0'))
    dt_s, choice = t(lambda: plan.auto_schedule(m.dst_dims))
    print(f'{name}: n_s {mf.n_s}: host->device + COO->CSR {dt_c:.3f} s, '
          f'auto_schedule {dt_s:.3f} s -> {choice["family"]}')
