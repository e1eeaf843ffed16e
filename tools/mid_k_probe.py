#!/usr/bin/env python3
"""Few to middling field counts (K = 4 ... 63) on the mappings scheduled as
LDS patches (bilinear, coarse -> fine): us per launch replayed from a hipGraph
-- the library's own choice, the LDS patch kernel forced, the lane-per-(row,
k) kernel forced.  What `remap_spmm.hip: patch_serves` rests on.  GPU box
only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from long_rows_probe import replay_us
from pyremap_amd import engine, synthetic
dev = torch.device('cuda', 0)
for wl in ('config1_esmf', 'config2', 'config4'):
    m = synthetic.make_config(wl, device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, index_base=1, device=dev)
    print(wl, plan.auto_schedule(m.dst_dims).get('family'))
    short = plan._split[0] if plan._split else plan
    for K in (4, 8, 12, 16, 24, 32, 40, 48, 63):
        x = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
        y = engine.remap_tensor(short, m.dst_dims, x, [0], engine.MODE_FRACB)
        ref = y.clone()
        row = {'wl': wl, 'K': K}
        for tag, tune in (('warm', None), ('default', None), ('patch', [5]), ('rowlane', [2])):
            def run():
                engine.remap_tensor(short, m.dst_dims, x, [0], engine.MODE_FRACB, out=y, tune=tune)
            try:
                row[tag] = round(replay_us(run, calls=20, reps=3), 2)
                assert torch.equal(torch.nan_to_num(y, nan=-2.5), torch.nan_to_num(ref, nan=-2.5))
            except Exception as exc:
                row[tag] = str(exc)[:50]
        del row['warm']
        print(json.dumps(row), flush=True)
