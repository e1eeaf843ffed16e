#!/usr/bin/env python3
"""
Which clock do the kernels run at?  A series of launches of a workload, the
shader clock measured right behind every series (`remap_clock_probe`: one
wave, 100 x Delta s_memtime / Delta s_memrealtime MHz), after idle periods of
different lengths -- config 4's LDS patch kernel is instruction-issue-bound
and its time follows the clock (5.2 or 6.4 ms); config 3's row-group kernel
waits on memory and does not.  GPU box only.

    python tools/clock_state.py [config4 config3 ...]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    for name in (sys.argv[1:] or ['config4', 'config3']):
        cfg = synthetic.CONFIGS[name]
        m = synthetic.make_config(name, device=dev, locality='mesh')
        plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                              m.n_a, m.n_b, index_base=1,
                                              device=dev)
        sched = plan.auto_schedule(m.dst_dims)
        K = cfg['K']
        x = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
        y = torch.empty((m.n_b, K), device=dev, dtype=torch.float64)

        def series(n):
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                engine.remap_tensor(plan, None, x, [0], engine.MODE_FRACB,
                                    out=y)
            b.record()
            mhz = engine.clock_probe(dev)
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n, mhz()
        n = 12 if name in ('config4', 'config5', 'headline') else 100
        series(3)
        for idle in (0.0, 0.0, 0.05, 0.5, 2.0, 0.0, 0.0):
            time.sleep(idle)
            ms, mhz = series(n)
            print(json.dumps(dict(workload=name, family=sched['family'],
                                  idle_before_s=idle, launches=n,
                                  ms_per_launch=round(ms, 4),
                                  clock_mhz=round(mhz, 1))), flush=True)
        del x, y, plan
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
