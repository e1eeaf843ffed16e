import time, torch, sys
sys.path.insert(0, '/root/repo')
from pyremap_amd import engine, synthetic
dev = torch.device('cuda', 0)
m = synthetic.conservative_map(2000, (20, 30), 1, 6, seed=1, device=dev)
plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, device=dev)
x = torch.randn((m.n_a, 128), device=dev, dtype=torch.float64)
y = engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB)
torch.cuda.synchronize()
for name, fn in (('remap_tensor', lambda: engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB, out=y)),
                 ('apply_strided', lambda: engine.apply_strided(plan, x, y.view(m.n_b, 128), n_batch=1, k_inner=128, x_row_stride=128, x_batch_stride=0, y_row_stride=128, y_batch_stride=0, mode=engine.MODE_FRACB))):
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, 'host us/call', (t1 - t0) / 2000 * 1e6, 'total us/call', (t2 - t0) / 2000 * 1e6)
