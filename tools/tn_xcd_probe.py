#!/usr/bin/env python3
"""
(Time, nCells) on config 3's map: does it pay to deal the patches of the
lanes-across-rows plan to the XCDs in COMPACT 2-D blocks (a 128-byte line of
a mesh-numbered time slice is then wanted by 1.34 XCDs on average instead of
1.51 with the contiguous ranges of the row-major patch list)?  The patch plan
is permuted in torch (patch-major arrays: cell lists, slots, entries), the
kernel is the product's.  GPU box only.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def permute_patches(q, perm):
    """The patch plan `q` with patch perm[i] at position i (the last, partial
    patch stays last)."""
    dev = q['ptr'].device
    rows = q['rows']
    n_p = q['n']
    ptr = q['ptr'].to(torch.int64)
    lens = (ptr[1:] - ptr[:-1])[perm]
    new_ptr = torch.zeros(n_p + 1, dtype=torch.int64, device=dev)
    new_ptr[1:] = torch.cumsum(lens, 0)
    start = torch.repeat_interleave(ptr[:-1][perm], lens)
    within = torch.arange(int(new_ptr[-1]), device=dev) - \
        torch.repeat_interleave(new_ptr[:-1], lens)
    ucol = q['ucol'][start + within]
    # slots
    n_slots = q['rowptr'].numel() - 1
    slot_old = (perm[:, None] * rows +
                torch.arange(rows, device=dev)[None, :]).reshape(-1)
    slot_old = slot_old[slot_old < n_slots]
    assert slot_old.numel() == n_slots
    rp = q['rowptr'].to(torch.int64)
    elen = (rp[1:] - rp[:-1])[slot_old]
    new_rp = torch.zeros(n_slots + 1, dtype=torch.int64, device=dev)
    new_rp[1:] = torch.cumsum(elen, 0)
    estart = torch.repeat_interleave(rp[:-1][slot_old], elen)
    ewithin = torch.arange(int(new_rp[-1]), device=dev) - \
        torch.repeat_interleave(new_rp[:-1], elen)
    eidx = estart + ewithin
    out = dict(q)
    out.update(ptr=new_ptr.to(torch.int32), ucol=ucol.contiguous(),
               rowptr=new_rp.to(torch.int32),
               lidx=q['lidx'][eidx].contiguous(),
               val=q['val'][eidx].contiguous(),
               order=q['order'][slot_old].contiguous())
    return out


def main():
    dev = torch.device('cuda', 0)
    m = synthetic.make_config('config3', device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    plan.auto_schedule(m.dst_dims)
    q = plan.cell_patches()
    ty, tx = q['tile']
    ny, nx = m.dst_dims
    npy, npx = (ny + ty - 1) // ty, (nx + tx - 1) // tx
    n_p = q['n']
    print('tile', q['tile'], 'patches', n_p, npy, npx, 'rows', q['rows'])
    ids = torch.arange(n_p, device=dev)
    yy, xx = ids // npx, ids % npx
    variants = {'row-major (shipped)': None}
    for by, bx in ((2, 4), (4, 2)):
        block = (yy * by // npy) * bx + (xx * bx // npx)
        key = block * n_p + ids
        if n_p * q['rows'] != m.n_b:          # the partial patch stays last
            key[-1] = 8 * n_p + n_p
        variants[f'compact {by}x{bx}'] = torch.argsort(key)
    T = 120
    xs = [torch.randn((T, m.n_a), device=dev, dtype=torch.float64)
          for _ in range(3)]
    ys = [torch.empty((T,) + tuple(m.dst_dims), device=dev,
                      dtype=torch.float64) for _ in range(3)]
    by = plan.algorithmic_bytes(T, 8, engine.MODE_FRACB)
    ref = None
    for rnd in range(2):
        for tag, perm in variants.items():
            plan._cell = q if perm is None else permute_patches(q, perm)
            plan._sched_version += 1

            def run(i):
                engine.remap_tensor(plan, m.dst_dims, xs[i % 3], [1],
                                    engine.MODE_FRACB, out=ys[i % 3])
            for i in range(6):
                run(i)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(60):
                run(i)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 60
            if ref is None:
                ref = ys[0].clone()
            same = torch.equal(torch.nan_to_num(ys[0], nan=-3.0),
                               torch.nan_to_num(ref, nan=-3.0))
            print(f'{tag:22s} {ms:.4f} ms  {by / (ms * 1e-3) / 8e12:.4f} '
                  f'of 8 TB/s  same bits: {same}')


if __name__ == '__main__':
    main()
