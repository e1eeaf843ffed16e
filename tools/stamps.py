#!/usr/bin/env python3
"""
Where does a wave of the default kernel spend a row?  (GPU box only.)

Build the diagnostic library first (in the build container):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared \
      -DREMAP_STAMPS -Iinclude -Ipyremap_amd/csrc \
      -o tools/_build/libremap_hip_stamps.so pyremap_amd/csrc/*.hip
then on the GPU box:
    REMAP_HIP_LIB=tools/_build/libremap_hip_stamps.so python tools/stamps.py

Shares only: the stamps serialise the phases (each one drains the counters),
so the run time of this build means nothing.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'config3'
tune = [int(t) for t in sys.argv[2].split(',')] if len(sys.argv) > 2 else \
    [6, 2, 1, 4, 2]
dev = torch.device('cuda', 0)
cfg = synthetic.CONFIGS[workload]
m = synthetic.make_config(workload, device=dev)
plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                      m.n_b, device=dev)
K = cfg['K']
x = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
y = torch.empty((m.n_b, K), device=dev, dtype=torch.float64)
for rep in range(3):
    buf = torch.zeros(8, dtype=torch.int64, device=dev)
    engine.apply_strided(plan, x, y, n_batch=1, k_inner=K, x_row_stride=K,
                         x_batch_stride=0, y_row_stride=K, y_batch_stride=0,
                         mode=engine.MODE_FRACB, tune=tune,
                         mask_out=buf.view(torch.uint8))
    torch.cuda.synchronize()
rows, p1, p2, p3, p4 = buf.tolist()[:5]
total = p1 + p2 + p3 + p4
print(f'{workload} tune {tune}: {rows} (row, chunk) units')
for name, v in (('row pointers (scalar trip)', p1),
                ('entries col/S (scalar trip)', p2),
                ('X loads issue + arrival (vector trip)', p3),
                ('accumulate + divide + issue stores', p4)):
    print(f'  {name:<42} {v / rows:9.0f} cycles/unit  {100 * v / total:5.1f} %')
print(f'  total {total / rows:.0f} cycles per unit (s_memtime ticks)')
