"""
CPU tests of the strip schedule (pyremap_amd/strips.py, kernel family 8):
the schedule is REPLAYED step by step the way csrc/spmm_strip.h executes it
-- arrivals issued `depth` steps ahead into ring slots, rows computed from
whatever the ring holds at that moment -- and must reproduce the CSR matrix,
entry by entry in CSR order.  (The kernel itself: tests/test_gpu_strips.py.)
"""
import types

import numpy as np
import pytest
import torch

from pyremap_amd import strips, synthetic


def _plan_like(m):
    """What build_strips reads of a RemapPlan, from a synthetic map on the
    CPU (scipy's COO -> CSR: remap_numpy.py:134-137)."""
    import scipy.sparse as sp
    mm = m.numpy()
    A = sp.csr_matrix((mm['S'], (mm['row'] - 1, mm['col'] - 1)),
                      shape=(m.n_b, m.n_a))
    A.sum_duplicates()
    A.sort_indices()
    return types.SimpleNamespace(
        n_a=m.n_a, n_b=m.n_b, device=torch.device('cpu'),
        rowptr=torch.from_numpy(A.indptr.astype(np.int64)),
        col=torch.from_numpy(A.indices.astype(np.int32)),
        val=torch.from_numpy(A.data.astype(np.float64)),
        frac_b=torch.from_numpy(mm['frac_b'])), A


def replay(st, A):
    """Execute the schedule as the kernel does; returns the number of
    (row, entry) pairs checked."""
    cap, spu, rpw, D, NW = (st['ring_slots'], st['steps_per_unit'],
                            st['rows_per_wave'], st['depth'], st['waves'])
    rps = NW * rpw
    arr_ptr = st['arr_ptr'].numpy()
    arr_src = st['arr_src'].numpy()
    arr_slot = st['arr_slot'].numpy()
    mptr = st['meta_ptr'].numpy()
    meta = st['meta'].numpy()                  # (16-byte units, 2) int64
    assert (np.diff(mptr) * 16 <= st['meta_slot_bytes']).all()
    assert st['meta_slot_bytes'] % 1024 == 0 and cap % 2 == 0
    assert (cap + 2) * strips.PIECE_BYTES + \
        (D + 1) * st['meta_slot_bytes'] == st['lds_bytes']
    zero = cap * strips.PIECE_BYTES
    seen = np.zeros(A.shape[0], dtype=np.int64)
    checked = 0
    for u in range(st['n_units']):
        T = int(st['unit_steps'][u])
        g0 = u * spu
        slots = np.full(cap, -1, dtype=np.int64)

        def issue(s):
            for i in range(arr_ptr[g0 + s], arr_ptr[g0 + s + 1]):
                assert 0 <= arr_slot[i] < cap // 2
                for h in (0, 1):
                    if arr_src[2 * i + h] >= 0:
                        slots[2 * arr_slot[i] + h] = arr_src[2 * i + h]
        for j in range(min(D, T)):
            issue(j)
        for t in range(T):
            # the loaders issue step t + D while step t is computed: the
            # worst case for a slot reused too early is "before"
            if t + D < T:
                issue(t + D)
            block = meta[mptr[g0 + t]:mptr[g0 + t + 1]]
            hdr, recs = block[:rps * 2], block[rps * 2:]
            for k in range(rps):
                r = int(np.int32(hdr[2 * k, 0] & 0xffffffff))
                e0 = int(hdr[2 * k, 0] >> 32)
                n = int(hdr[2 * k, 1])
                if r < 0:
                    assert n == 0
                    continue
                seen[r] += 1
                lo, hi = A.indptr[r], A.indptr[r + 1]
                real = hi - lo
                assert n % strips.QUANT == 0 and 0 <= n - real < strips.QUANT
                assert hdr[2 * k + 1, 0:1].view(np.float64)[0] == \
                    st['frac_b'][r]
                off = recs[e0:e0 + n, 0]
                wts = recs[e0:e0 + n, 1].copy().view(np.float64)
                assert (off[:real] % strips.PIECE_BYTES == 0).all()
                assert np.array_equal(
                    slots[off[:real] // strips.PIECE_BYTES],
                    A.indices[lo:hi]), (u, t, r)
                assert np.array_equal(wts[:real], A.data[lo:hi])
                # the pads: +0.0 on the slot of zeros
                assert (off[real:] == zero).all()
                assert (recs[e0 + real:e0 + n, 1] == 0).all()
                checked += real
    assert (seen == 1).all()
    return checked


@pytest.mark.parametrize('shape', [
    dict(strip_rows=8, step_cols=2, segments=3, depth=2, waves=8),
    dict(strip_rows=14, step_cols=1, segments=1, depth=2, waves=14),
    dict(strip_rows=16, step_cols=1, segments=1, depth=1, waves=6, gap=0),
    dict(strip_rows=4, step_cols=4, segments=2, depth=3, waves=13, gap=2),
])
@pytest.mark.parametrize('locality', ['mesh', 'raster'])
def test_schedule_replays_to_the_csr_matrix(shape, locality):
    # grid dims that the strip / step / segment sizes do not divide
    m = synthetic.conservative_map(1200, (37, 61), 6, 20, seed=3,
                                   signed=True, locality=locality)
    plan, A = _plan_like(m)
    st = strips.build_strips(plan, m.dst_dims, **shape)
    st['frac_b'] = plan.frac_b.numpy()
    assert replay(st, A) == A.nnz
    # what neighbouring steps share is fetched once: far fewer arrivals than
    # (step, source row) needs, and no more than a few per source row
    assert st['arrivals'] < 0.5 * A.nnz
    assert st['lds_bytes'] <= strips.LDS_BYTES


def test_a_ring_that_cannot_fit_is_refused():
    m = synthetic.conservative_map(1200, (37, 61), 6, 20, seed=3,
                                   locality='scatter')
    plan, _ = _plan_like(m)
    with pytest.raises(strips.StripsUnfit):
        strips.build_strips(plan, m.dst_dims, strip_rows=8, step_cols=2,
                            max_bytes=16 * 1024)


def test_rows_dealt_evenly_to_the_waves():
    m = synthetic.conservative_map(1500, (32, 64), 6, 30, seed=9)
    plan, A = _plan_like(m)
    st = strips.build_strips(plan, m.dst_dims, strip_rows=16, step_cols=2,
                             waves=8, max_bytes=1 << 30)
    rpw, NW = st['rows_per_wave'], st['waves']
    rps = NW * rpw
    meta, mptr = st['meta'], st['meta_ptr']
    hdr_rows = (mptr[:-1, None] + torch.arange(rps)[None, :] * 2).reshape(-1)
    lens = meta[hdr_rows, 1].reshape(-1, NW, rpw)
    per_wave = lens.sum(2)
    busy = per_wave.sum(1) > 0
    spread = (per_wave.max(1).values - per_wave.min(1).values)[busy]
    # longest-first, dealt back and forth: the waves of a step differ by
    # less than one long row
    assert int(spread.max()) <= 32


def test_free_list_needs_far_fewer_slots_than_a_fifo():
    """Slot pairs are reused as soon as their pieces are dead: the LDS holds
    about what is alive (pieces read by the current step + those in flight),
    not everything between the oldest live arrival and the newest."""
    m = synthetic.conservative_map(1200, (37, 61), 6, 20, seed=3,
                                   locality='mesh')
    plan, A = _plan_like(m)
    st = strips.build_strips(plan, m.dst_dims, strip_rows=8, step_cols=1,
                             segments=1, depth=2, waves=8)
    st['frac_b'] = plan.frac_b.numpy()
    assert replay(st, A) == A.nnz
    # pieces one step reads, at most
    rps = st['waves'] * st['rows_per_wave']
    meta, mptr = st['meta'].numpy(), st['meta_ptr'].numpy()
    most = 0
    for g in range(len(mptr) - 1):
        recs = meta[mptr[g] + rps * 2:mptr[g + 1], 0]
        most = max(most, len(np.unique(recs)))
    assert st['ring_slots'] <= 2.2 * most


def test_gap_keeps_halo_pieces_resident():
    """A piece may sit out `gap` steps that do not read it: fewer arrivals
    (less HBM traffic), a few more ring slots -- the same matrix."""
    m = synthetic.conservative_map(1200, (37, 61), 6, 20, seed=3,
                                   locality='mesh')
    plan, A = _plan_like(m)
    out = []
    for gap in (0, 1, 3):
        st = strips.build_strips(plan, m.dst_dims, strip_rows=8,
                                 step_cols=1, segments=1, depth=2, gap=gap,
                                 waves=8, max_bytes=1 << 30)
        st['frac_b'] = plan.frac_b.numpy()
        assert replay(st, A) == A.nnz
        out.append((st['arrivals'], st['ring_slots']))
    assert out[0][0] > out[1][0] > out[2][0]
    assert max(o[1] for o in out) <= 1.3 * min(o[1] for o in out)
