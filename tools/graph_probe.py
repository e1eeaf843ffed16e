#!/usr/bin/env python3
"""
Launch-bound calls replayed from a hipGraph: `remap_apply_f64` launches on the
caller's stream and neither allocates nor synchronises, so a series of calls
can be captured once (torch.cuda.CUDAGraph) and replayed.  Per-call device
time, launched one by one from Python against the same calls replayed from a
graph (GPU box only).

    python tools/graph_probe.py [--workload config3] [--calls 50]
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
    ap.add_argument('--calls', type=int, default=50)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    m = synthetic.make_config(args.workload, device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    plan.auto_schedule(m.dst_dims)
    for shape, axis in (((m.n_a,), 0), ((1, m.n_a), 1), ((12, m.n_a), 1),
                        ((m.n_a, 8), 0), ((m.n_a, 64), 0)):
        n = args.calls
        xs = [torch.randn(shape, device=dev, dtype=torch.float64)
              for _ in range(n)]
        for mode, name in ((engine.MODE_FRACB, 'fracb'),
                           (engine.MODE_MASKED, 'masked')):
            outs = [engine.remap_tensor(plan, m.dst_dims, x, [axis], mode,
                                        threshold=0.01) for x in xs]
            want = [o.clone() for o in outs]

            def calls():
                for x, o in zip(xs, outs):
                    engine.remap_tensor(plan, m.dst_dims, x, [axis], mode,
                                        threshold=0.01, out=o)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            calls()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a.record()
            for _ in range(args.reps):
                calls()
            b.record()
            host = (time.perf_counter() - t0) / (args.reps * n)
            torch.cuda.synchronize()
            eager = a.elapsed_time(b) / (args.reps * n)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                calls()
            for o in outs:
                o.fill_(-1.0)
            graph.replay()
            torch.cuda.synchronize()
            same = all(bool(((o == w) | (o.isnan() & w.isnan())).all())
                       for o, w in zip(outs, want))
            a.record()
            for _ in range(args.reps):
                graph.replay()
            b.record()
            torch.cuda.synchronize()
            replay = a.elapsed_time(b) / (args.reps * n)
            by = plan.algorithmic_bytes(xs[0].numel() // m.n_a, 8, mode)
            print(f'{str(shape):16s} {name:7s} one by one {eager * 1e3:7.2f} '
                  f'us (host {host * 1e6:6.2f} us per call)   graph replay '
                  f'{replay * 1e3:7.2f} us = {by / replay / 1e6 / 8000:.3f} '
                  f'of 8 TB/s   {"bitwise" if same else "DIFFERS"}')


if __name__ == '__main__':
    main()
