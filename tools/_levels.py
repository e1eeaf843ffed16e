import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyremap_amd import engine, synthetic
dev = torch.device('cuda', 0)
m = synthetic.make_config('config3', device=dev)
plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, device=dev)
print(plan.auto_schedule(m.dst_dims))
for T, L in ((8, 64), (8, 60), (8, 61), (10, 48), (8, 56), (5, 100), (8, 72), (8, 50), (9, 57)):
    xs = [torch.randn((T, m.n_a, L), device=dev, dtype=torch.float64) for _ in range(3)]
    for mode, name in ((engine.MODE_FRACB, 'fracb'),):
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
