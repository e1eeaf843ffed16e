#!/usr/bin/env python3
"""
(Time, nCells, nVertLevels) fields addressed in place on config 3's mapping,
for the level counts real model output has (GPU box only): ms per launch,
algorithmic GB/s and fraction of the 8 TB/s roofline per (Time, levels).

    python tools/levels_sweep.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyremap_amd import engine, synthetic
dev = torch.device('cuda', 0)
m = synthetic.make_config('config3', device=dev)
plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, device=dev)
print(plan.auto_schedule(m.dst_dims))
for T, L in ((8, 64), (8, 60), (8, 61), (8, 80), (5, 100), (4, 128), (4, 137),
             (16, 32), (1, 60), (1, 100)):
    xs = [torch.randn((T, m.n_a, L), device=dev, dtype=torch.float64) for _ in range(3)]
    for mode, name in ((engine.MODE_FRACB, 'fracb'), (engine.MODE_MASKED, 'masked')):
        if mode == engine.MODE_MASKED:
            for x in xs:
                x[:, torch.rand(m.n_a, device=dev) < 0.25, L // 2:] = float('nan')
        outs = [torch.empty((T,) + tuple(m.dst_dims) + (L,), device=dev, dtype=torch.float64) for _ in range(3)]
        for i in range(10):
            engine.remap_tensor(plan, m.dst_dims, xs[i % 3], [1], mode, threshold=0.01, out=outs[i % 3])
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        n = 60
        for i in range(n):
            engine.remap_tensor(plan, m.dst_dims, xs[i % 3], [1], mode, threshold=0.01, out=outs[i % 3])
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        K = T * L
        by = plan.algorithmic_bytes(K, 8, mode)
        print(f'T={T:3d} L={L:4d} K={K:4d} {name:7s} {ms:8.4f} ms  {by / ms / 1e6:8.1f} GB/s  frac {by / ms / 1e6 / 8000:.3f}')
