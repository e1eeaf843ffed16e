#!/usr/bin/env python3
"""
Pole-capped bilinear map, K = 64: the two launches of a split plan issued
eagerly, with and without hipExtAnyOrderLaunch on the long rows' launch
(diagnostic build, REMAP_ANY_ORDER=1).  GPU box only.

    REMAP_HIP_LIB=tools/_build/libremap_hip_diag.so [REMAP_ANY_ORDER=1] \
        python tools/any_order_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    m = synthetic.make_config('config1_esmf', device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    plan.auto_schedule(m.dst_dims)
    for K in (24, 64, 128):
        x = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
        y = engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB)
        ref = y.clone()
        times = []
        for _ in range(5):
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200):
                engine.remap_tensor(plan, m.dst_dims, x, [0],
                                    engine.MODE_FRACB, out=y)
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / 200 * 1e3)
        assert torch.equal(y, ref)
        print(f'K={K}: {min(times):.1f} us per apply (eager), any-order '
              f'{"on" if os.environ.get("REMAP_ANY_ORDER") else "off"}')


if __name__ == '__main__':
    main()
