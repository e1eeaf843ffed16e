#!/usr/bin/env python3
"""Few to middling field counts (K = 4 ... 63): us per launch replayed from a
hipGraph -- the library's own choice, the mapping's scheduled family forced
(LDS patches on the bilinear / coarse -> fine maps, row groups on the
conservative ones), the scalar-cache rows and the lane-per-(row, k) kernel
forced.  What `remap_spmm.hip: patch_serves` and the row groups' rule in
`hint_usable` rest on.  GPU box only.

    python tools/mid_k_probe.py [f32] [masked] [workload ...]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from long_rows_probe import replay_us  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    argv = sys.argv[1:]
    dtype = torch.float32 if 'f32' in argv else torch.float64
    mode = engine.MODE_MASKED if 'masked' in argv else engine.MODE_FRACB
    workloads = [a for a in argv if a not in ('f32', 'masked')] or \
        ['config1_esmf', 'config2', 'config4', 'config3', 'headline',
         'config5']
    for wl in workloads:
        m = synthetic.make_config(wl, device=dev, locality='mesh')
        plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                              m.n_a, m.n_b, index_base=1,
                                              device=dev)
        choice = plan.auto_schedule(m.dst_dims)
        part = plan._split[0] if plan._split else plan
        hint = part.default_tune
        if isinstance(hint, dict):
            hint = hint.get(mode)
        scheduled = list(hint) if hint else \
            ([5] if part.patches is not None else [6])
        print(wl, choice.get('family'), scheduled, flush=True)
        for K in (4, 8, 12, 16, 24, 32, 40, 48, 63):
            x = torch.randn((m.n_a, K), device=dev,
                            dtype=torch.float64).to(dtype)
            if mode == engine.MODE_MASKED:
                x[torch.rand(m.n_a, device=dev) < 0.2] = float('nan')
            y = engine.remap_tensor(part, m.dst_dims, x, [0], mode,
                                    threshold=0.1)
            ref = y.clone()
            row = {'wl': wl, 'K': K}
            # (the first variant is timed twice: the clock ramps)
            for tag, tune in (('warm', None), ('default', None),
                              ('scheduled', scheduled), ('rowscalar', [6]),
                              ('rowlane', [2])):
                def run():
                    engine.remap_tensor(part, m.dst_dims, x, [0], mode,
                                        threshold=0.1, out=y, tune=tune)
                try:
                    row[tag] = round(replay_us(run, calls=10, reps=3), 2)
                    assert torch.equal(torch.nan_to_num(y, nan=-2.5),
                                       torch.nan_to_num(ref, nan=-2.5))
                except Exception as exc:   # noqa: BLE001
                    row[tag] = str(exc)[:50]
            del row['warm']
            print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
