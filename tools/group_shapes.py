#!/usr/bin/env python3
"""union entries / entries for candidate row-group shapes (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'config5'
dev = torch.device('cuda', 0)
m = synthetic.make_config(name, device=dev)
plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                      m.n_b, device=dev)
my, mx = m.dst_dims
lens = plan.rowptr[1:] - plan.rowptr[:-1]
rows = torch.repeat_interleave(torch.arange(m.n_b, device=dev), lens)
jy, jx = rows // mx, rows % mx
col = plan.col[:plan.nnz].to(torch.int64)
for gy, gx in ((1, 8), (2, 4), (2, 8), (4, 4), (4, 8), (8, 8)):
    g = (jy // gy) * ((mx + gx - 1) // gx) + jx // gx
    nu = torch.unique(g * m.n_a + col).numel()
    print(f'{name}: group {gy}x{gx} ({gy * gx:2d} rows): union/nnz = '
          f'{nu / plan.nnz:.3f}  loads per row = {nu / m.n_b:.2f}')
