"""
The N > 1 path on CPU: several processes, gloo backend.  What is exercised is
the distributed plumbing of pyremap_amd.parallel -- work-balanced row
sharding (uneven shards included), the exchange of the source field (one
broadcast, or each rank receiving only the PACKED source rows its shard
references: one all_to_all_single), pipelined batches, the row gather, and
the zero-collective field-sharded mode -- with the CPU oracle standing in for
the HIP kernel as the per-rank compute (the kernel itself is covered by the
-m gpu tests; shards there are checked in test_gpu_parity.py and
test_gpu_multi.py).

The mapping's source cells are numbered the way an MPAS mesh numbers them
(`synthetic.mesh_numbering`), NOT along the destination raster: the packed
exchange must move about (1/N + halo) of the field all the same, where a
(min, max) band of source rows -- what round 2 sent -- is the whole field.
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


def _problem():
    from oracle import oracle
    from pyremap_amd import synthetic
    m = synthetic.conservative_map(900, (20, 30), 1, 6, seed=5,
                                   locality='mesh')
    mm = m.numpy()
    # a solid band of land (rows 0-199 empty): shards of equal WORK then
    # hold unequal numbers of rows
    keep = mm['row'] > 200
    for key in ('row', 'col', 'S'):
        mm[key] = mm[key][keep]
    mm['frac_b'][:200] = 0.0
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    return m, mm, csr


def _shard(csr, r0, r1, n_a):
    from oracle import oracle
    return oracle.OracleCSR(csr.indptr[r0:r1 + 1] - csr.indptr[r0],
                            csr.indices[csr.indptr[r0]:csr.indptr[r1]],
                            csr.data[csr.indptr[r0]:csr.indptr[r1]],
                            (r1 - r0, n_a))


def _worker(rank, world, port, tmpdir, how):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle
        from pyremap_amd import parallel
        m, mm, csr = _problem()
        K = 24
        rng = np.random.default_rng(0)
        full_x = rng.standard_normal((m.n_a, K))
        ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], full_x, False,
                                          0.0)
        ref[ref_mask] = np.nan
        notes = {}
        if how == 'fields':
            # zero-collective mode: replicated weights, K split over ranks,
            # every rank loads its own fields -- no communication at all
            k0, k1 = rank * K // world, (rank + 1) * K // world
            y, mask = oracle.remap_flat(csr, mm['frac_b'],
                                        full_x[:, k0:k1], False, 0.0)
            y[mask] = np.nan
            ok = np.array_equal(y, ref[:, k0:k1], equal_nan=True)
            r0 = r1 = 0
        else:
            bounds = parallel.row_shard_bounds(
                torch.from_numpy(csr.indptr), world)
            r0, r1 = bounds[rank], bounds[rank + 1]
            shard = _shard(csr, r0, r1, m.n_a)
            # only rank 0 has the field; everyone allocates the buffer, and
            # poisons it so a row that never arrived cannot go unnoticed
            x = torch.full((m.n_a, K), float('nan'), dtype=torch.float64)
            if rank == 0:
                x = torch.from_numpy(full_x.copy())
            if how == 'broadcast':
                parallel.broadcast_field(x, src=0)
                x_use, shard_use = x, shard
            else:
                # packed columns: the shard in the compact space of the
                # source rows it references (monotone renumbering: the order
                # of a row's entries -- the order of the sums -- is kept)
                ucols = [parallel.unique_columns(torch.from_numpy(
                    _shard(csr, bounds[r], bounds[r + 1], m.n_a).indices))
                    for r in range(world)]
                mine = ucols[rank]
                packed_idx = np.searchsorted(mine.numpy(), shard.indices)
                shard_use = oracle.OracleCSR(
                    shard.indptr, packed_idx.astype(np.int32), shard.data,
                    (r1 - r0, len(mine)))
                counts = [len(u) for u in ucols]
                notes['packed_frac'] = parallel.packed_fraction(counts,
                                                                m.n_a)
                notes['band_frac'] = sum(
                    int(u.max()) + 1 - int(u.min()) for u in ucols
                    if len(u)) / (world * m.n_a)

                def pieces_of(t):
                    return [t[u].contiguous() for u in ucols] \
                        if rank == 0 else None
                if how == 'packed':
                    recv, _ = parallel.scatter_packed(
                        pieces_of(x), len(mine) * K, torch.float64,
                        torch.device('cpu'), src=0)
                else:   # 'packed_async': two batches, second one in flight
                    x2 = torch.from_numpy(2.0 * full_x) if rank == 0 \
                        else None
                    recv, w1 = parallel.scatter_packed(
                        pieces_of(x), len(mine) * K, torch.float64,
                        torch.device('cpu'), src=0, async_op=True)
                    recv2, w2 = parallel.scatter_packed(
                        pieces_of(x2), len(mine) * K, torch.float64,
                        torch.device('cpu'), src=0, async_op=True)
                    w1.wait()
                    w2.wait()
                    y2, mask2 = oracle.remap_flat(
                        shard_use, mm['frac_b'][r0:r1],
                        recv2.reshape(len(mine), K).numpy(), False, 0.0)
                    y2[mask2] = np.nan
                    notes['second'] = int(np.array_equal(
                        y2, 2.0 * ref[r0:r1], equal_nan=True))
                x_use = recv.reshape(len(mine), K)
                notes['received_rows'] = len(mine)
            x, shard = x_use, shard_use
            y_local, mask = oracle.remap_flat(shard, mm['frac_b'][r0:r1],
                                              x.numpy(), False, 0.0)
            y_local[mask] = np.nan
            y = parallel.gather_rows(torch.from_numpy(y_local), bounds)
            ok = np.array_equal(y.numpy(), ref, equal_nan=True)
            work = [int(csr.indptr[bounds[i + 1]] - csr.indptr[bounds[i]]) +
                    2 * (bounds[i + 1] - bounds[i]) for i in range(world)]
            notes['balanced'] = int(max(work) <=
                                    1.15 * sum(work) / world + 20)
            notes['uneven'] = int(
                max(bounds[i + 1] - bounds[i] for i in range(world)) >
                1.1 * m.n_b / world)
        with open(os.path.join(tmpdir, f'rank{rank}.txt'), 'w') as f:
            f.write(repr(dict(ok=int(ok), r0=r0, r1=r1, **notes)))
    finally:
        dist.destroy_process_group()


def _run(tmp_path, world, how):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), how),
             nprocs=world, join=True)
    out = [eval(open(tmp_path / f'rank{rank}.txt').read())
           for rank in range(world)]
    for rank, o in enumerate(out):
        assert o['ok'] == 1, f'rank {rank} ({how}): result differs'
    return out


def test_row_sharded_remap_world_size_2(tmp_path):
    out = _run(tmp_path, 2, 'broadcast')
    assert all(o['balanced'] == 1 for o in out)
    assert out[0]['r0'] == 0 and out[0]['r1'] == out[1]['r0']
    assert out[1]['r1'] == 600


def test_row_sharded_world_size_4_uneven_shards_packed(tmp_path):
    """Four ranks, shards of equal work but unequal row counts, and each rank
    receiving only the packed source rows it references -- on a source mesh
    whose numbering has nothing to do with the destination raster."""
    out = _run(tmp_path, 4, 'packed')
    assert all(o['balanced'] == 1 for o in out)
    assert any(o['uneven'] == 1 for o in out)
    spans = [(o['r0'], o['r1']) for o in out]
    assert spans[0][0] == 0 and spans[-1][1] == 600
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # the packed rows are about (1/N + halo) of the field per rank ...
    assert out[0]['packed_frac'] < 0.55
    assert sum(o['received_rows'] for o in out) < 0.55 * 4 * 900
    # ... where a (min, max) band of source rows is (nearly) all of it
    assert out[0]['band_frac'] > 0.9


def test_row_sharded_pipelined_batches(tmp_path):
    out = _run(tmp_path, 3, 'packed_async')
    assert all(o['second'] == 1 for o in out)


def test_field_sharded_needs_no_collective(tmp_path):
    _run(tmp_path, 4, 'fields')
