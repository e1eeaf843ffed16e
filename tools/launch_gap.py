#!/usr/bin/env python3
"""
How much of a small (1/8-shard) step is not kernel time?  Times the same
launches (a) with an event pair per step, (b) with one event pair around the
region, (c) replayed from a hipGraph.   GPU box only.

    python tools/launch_gap.py --shard 0/8
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config3')
    ap.add_argument('--shard', default='0/8')
    ap.add_argument('--steps', type=int, default=400)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = synthetic.CONFIGS[args.workload]
    K = cfg['K']
    m = synthetic.make_config(args.workload, device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    r, n = (int(v) for v in args.shard.split('/'))
    if n > 1:
        plan = plan.shard(r, n)
    print(plan.auto_schedule(m.dst_dims))
    xs = [torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
          for _ in range(2)]
    ys = [torch.empty((plan.n_b, K), device=dev, dtype=torch.float64)
          for _ in range(2)]

    def launch(i):
        engine.remap_tensor(plan, None, xs[i % 2], [0], engine.MODE_FRACB,
                            out=ys[i % 2])

    for i in range(20):
        launch(i)
    torch.cuda.synchronize()
    steps = args.steps

    def wall(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / steps

    def per_step_events():
        ev = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

        def run():
            for i in range(steps):
                ev[i][0].record()
                launch(i)
                ev[i][1].record()
        w = wall(run)
        k = sum(a.elapsed_time(b) for a, b in ev) / steps
        return w, k

    def region_events():
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)

        def run():
            a.record()
            for i in range(steps):
                launch(i)
            b.record()
        w = wall(run)
        return w, a.elapsed_time(b) / steps

    def graph():
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            launch(0)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for i in range(steps):
                    launch(i)
        g.replay()
        torch.cuda.synchronize()
        return wall(g.replay), None

    for rnd in range(3):
        w1, k1 = per_step_events()
        w2, k2 = region_events()
        w3, _ = graph()
        print(f'round {rnd}: per-step events wall {w1 * 1e3:.1f} us '
              f'(events mean {k1 * 1e3:.1f}); region events wall '
              f'{w2 * 1e3:.1f} us (events {k2 * 1e3:.1f}); graph wall '
              f'{w3 * 1e3:.1f} us')


if __name__ == '__main__':
    main()
