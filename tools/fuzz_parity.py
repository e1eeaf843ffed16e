#!/usr/bin/env python3
"""
Time-bounded fuzz of the GPU parity suite's randomised end-to-end case
(tests/test_gpu_parity.py::test_random_cases_bitwise: random mapping, grid,
schedule, layout, dtype, mode, shard -- bit for bit against the oracle) over
seeds the suite itself does not run (GPU box only).

    python tools/fuzz_parity.py [seconds=540] [first_seed=12]

Round 2: seeds 12 ... 497 (4 860 cases and their shards) in 543 s, no
mismatch -- after seed 188 had shown the TEST wrapping a NaN-free field as a
MaskedArray, which the reference never does (remap_numpy.py:201-204).
"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))

import torch  # noqa: E402

import test_gpu_parity as t  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 540.0
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = torch.device('cuda', 0)
    t0 = time.time()
    n, bad, seed = 0, [], first
    while time.time() - t0 < budget and len(bad) < 4:
        try:
            t.test_random_cases_bitwise(dev, seed)
            n += 1
        except Exception as exc:   # noqa: BLE001 - reported
            print('SEED', seed, 'FAILED', type(exc).__name__, str(exc)[:300])
            bad.append(seed)
        seed += 1
    print(f'seeds ok: {n}, failed: {bad}, next seed {seed}, '
          f'{time.time() - t0:.0f} s')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
