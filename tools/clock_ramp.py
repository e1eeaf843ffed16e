#!/usr/bin/env python3
"""
Why does a short timed region (the driver's --steps 20 --warmup 5) average
slower than a long one?  (GPU box only.)

For idle gaps of 0 ... 1 s in front of it, run `warmup` untimed launches and
then `steps` launches of the config-3 kernel, three ways:
  * one HIP event pair around the whole region (what bench.py's mean is),
  * an event pair per launch (in order, so a ramp shows as a trend),
and print both.  A second table repeats it with the GPU kept busy by a
streaming copy right up to the first warm-up launch.
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
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--sets', type=int, default=3)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = synthetic.CONFIGS[args.workload]
    m = synthetic.make_config(args.workload, device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    print('schedule', plan.auto_schedule(m.dst_dims))
    K = cfg['K']
    xs = [torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
          for _ in range(args.sets)]
    ys = [torch.empty(tuple(m.dst_dims) + (K,), device=dev,
                      dtype=torch.float64) for _ in range(args.sets)]
    big_a = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    big_b = torch.empty(1 << 28, dtype=torch.uint8, device=dev)

    def launch(i):
        s = i % args.sets
        engine.remap_tensor(plan, m.dst_dims, xs[s], [0], engine.MODE_FRACB,
                            out=ys[s])

    for i in range(10):
        launch(i)
    torch.cuda.synchronize()

    def region(idle_s, busy):
        torch.cuda.synchronize()
        time.sleep(idle_s)
        if busy:
            for _ in range(int(busy)):   # 256 MiB copies, ~0.085 ms each
                engine.stream_copy(big_b, big_a)
        for i in range(args.warmup):
            launch(i)
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for i in range(args.steps):
            launch(args.warmup + i)
        b.record()
        torch.cuda.synchronize()
        whole = a.elapsed_time(b) / args.steps
        torch.cuda.synchronize()
        time.sleep(idle_s)
        if busy:
            for _ in range(int(busy)):
                engine.stream_copy(big_b, big_a)
        for i in range(args.warmup):
            launch(i)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
        for i in range(args.steps):
            ev[i][0].record()
            launch(args.warmup + i)
            ev[i][1].record()
        torch.cuda.synchronize()
        per = [x.elapsed_time(y) for x, y in ev]
        return whole, per

    for busy in (0, 80, 240, 800, 2400):
        print(f'--- {busy} x 256 MiB copies (~{busy * 0.085:.0f} ms busy) '
              f'between the idle gap and the warm-up; warmup {args.warmup}, '
              f'steps {args.steps}')
        for idle in ((0.0, 0.001, 0.01, 0.1, 1.0) if busy == 0 else
                     (0.1, 1.0)):
            for rep in range(2):
                whole, per = region(idle, busy)
                s = sorted(per)
                print(f'idle {idle:6.3f} s  region mean {whole:.4f} ms | '
                      f'per-launch mean {sum(per) / len(per):.4f} median '
                      f'{s[len(s) // 2]:.4f} min {s[0]:.4f} max {s[-1]:.4f} | '
                      f'first 8: ' + ' '.join(f'{t:.3f}' for t in per[:8]))
    # a long steady run for comparison
    for i in range(50):
        launch(i)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(300):
        launch(i)
    b.record()
    torch.cuda.synchronize()
    print(f'steady: 300 launches, mean {a.elapsed_time(b) / 300:.4f} ms')


if __name__ == '__main__':
    main()
