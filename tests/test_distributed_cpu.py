"""
The N > 1 path on CPU: two processes, gloo backend.  What is exercised is the
distributed plumbing of pyremap_amd.parallel -- work-balanced row sharding,
the single broadcast of the source field, the row gather -- with the CPU
oracle standing in for the HIP kernel as the per-rank compute (the kernel
itself is covered by the -m gpu tests; shards there are checked in
test_gpu_parity.py::test_row_range_and_shards).
"""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle
        from pyremap_amd import parallel, synthetic
        m = synthetic.conservative_map(900, (20, 30), 1, 6, seed=5)
        mm = m.numpy()
        csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'],
                                m.n_b, m.n_a)
        K = 24
        # only rank 0 has the field; everyone allocates the buffer
        x = torch.zeros((m.n_a, K), dtype=torch.float64)
        if rank == 0:
            x = torch.from_numpy(
                np.random.default_rng(0).standard_normal((m.n_a, K)))
        parallel.broadcast_field(x, src=0)
        bounds = parallel.row_shard_bounds(torch.from_numpy(csr.indptr),
                                           world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        shard = oracle.OracleCSR(csr.indptr[r0:r1 + 1] - csr.indptr[r0],
                                 csr.indices[csr.indptr[r0]:csr.indptr[r1]],
                                 csr.data[csr.indptr[r0]:csr.indptr[r1]],
                                 (r1 - r0, m.n_a))
        y_local, mask = oracle.remap_flat(shard, mm['frac_b'][r0:r1],
                                          x.numpy(), False, 0.0)
        y_local[mask] = np.nan
        y = parallel.gather_rows(torch.from_numpy(y_local), bounds)
        ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x.numpy(),
                                          False, 0.0)
        ref[ref_mask] = np.nan
        ok = np.array_equal(y.numpy(), ref, equal_nan=True)
        balanced = abs((r1 - r0) - m.n_b / world) < 0.35 * m.n_b
        with open(os.path.join(tmpdir, f'rank{rank}.txt'), 'w') as f:
            f.write(f'{int(ok)} {int(balanced)} {r0} {r1}')
    finally:
        dist.destroy_process_group()


def test_row_sharded_remap_world_size_2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)),
             nprocs=world, join=True)
    spans = []
    for rank in range(world):
        ok, balanced, r0, r1 = open(
            tmp_path / f'rank{rank}.txt').read().split()
        assert ok == '1', f'rank {rank}: gathered result differs'
        assert balanced == '1'
        spans.append((int(r0), int(r1)))
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0]
    assert spans[1][1] == 600
