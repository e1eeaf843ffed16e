#!/usr/bin/env python3
"""A mapping scheduled as LDS patches (family 5) -- by default the SHORT rows
of the pole-capped 1 deg -> 0.5 deg bilinear map, config1_esmf -- on patch
plans of several tile sizes: us per launch replayed from a hipGraph, (n, K)
fields.  GPU box only.

    python tools/short_rows_tile_probe.py [workload [K,K,...]]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from long_rows_probe import replay_us  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    workload = sys.argv[1] if len(sys.argv) > 1 else 'config1_esmf'
    ks = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 \
        else [64, 128, 512]
    m = synthetic.make_config(workload, device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    print(json.dumps(plan.auto_schedule(m.dst_dims), default=str))
    short = plan._split[0] if plan._split else plan
    print('shipped', short.patches['tile'], short.patches['umax'])
    for K in ks:
        x = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
        y = engine.remap_tensor(short, m.dst_dims, x, [0], engine.MODE_FRACB)
        ref = y.clone()
        row = {'K': K}
        for tile in ('shipped', (4, 8), (8, 8), (8, 16), (16, 16), (16, 32),
                     (24, 24), (32, 32)):
            if tile != 'shipped':
                short.build_patches(m.dst_dims, tile=tile,
                                    lds_budget=150 * 1024,
                                    row_bytes=512 if K <= 64 else 1024)

            def run():
                engine.remap_tensor(short, m.dst_dims, x, [0],
                                    engine.MODE_FRACB, out=y, tune=[5])
            us = replay_us(run)
            same = bool(torch.equal(torch.nan_to_num(y, nan=-2.5),
                                    torch.nan_to_num(ref, nan=-2.5)))
            row[str(tile if tile == 'shipped' else short.patches['tile'])] = \
                [round(us, 2), same]
        print(json.dumps(row), flush=True)
        short.auto_schedule(m.dst_dims, _split_ok=False)


if __name__ == '__main__':
    main()
