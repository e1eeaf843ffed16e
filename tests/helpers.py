"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def golden_files(pattern='g[01256]_*.npz'):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def assert_bitwise(actual, expected, what=''):
    """
    Bit-exact comparison for fp64 results: identical NaN placement, and
    identical bit patterns (sign of zero included) everywhere else.  NaN
    payloads are not compared (x86 and gfx950 pick different default NaNs).
    """
    actual = np.ascontiguousarray(actual, dtype=np.float64)
    expected = np.ascontiguousarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape, \
        f'{what}: shape {actual.shape} != {expected.shape}'
    nan_a = np.isnan(actual)
    nan_e = np.isnan(expected)
    assert np.array_equal(nan_a, nan_e), \
        f'{what}: NaN placement differs at {np.argwhere(nan_a != nan_e)[:5]}'
    bits_a = actual.view(np.int64)[~nan_a]
    bits_e = expected.view(np.int64)[~nan_e]
    bad = bits_a != bits_e
    if bad.any():
        idx = np.flatnonzero(bad)[:5]
        raise AssertionError(
            f'{what}: {bad.sum()} of {bad.size} values differ bitwise, e.g. '
            f'{actual[~nan_a][idx]} vs {expected[~nan_e][idx]}')


def golden_cases(path):
    """Yield (index, field-as-handed-over, remap_axes, thr, out, mask)."""
    g = np.load(path)
    for i in range(int(g['n_cases'])):
        field = g[f'c{i}_field']
        thr = float(g[f'c{i}_thr'])
        thr = None if np.isnan(thr) else thr
        if f'c{i}_in_mask' in g:
            arg = np.ma.masked_array(field, g[f'c{i}_in_mask'])
        elif bool(g[f'c{i}_was_masked_array']):
            arg = np.ma.masked_array(field, np.isnan(field))
        else:
            arg = field
        yield (i, arg, [int(a) for a in g[f'c{i}_remap_axes']], thr,
               g[f'c{i}_out'], g[f'c{i}_mask'])


def golden_map(path):
    g = np.load(path)
    return {k: g[k] for k in ('n_a', 'n_b', 'src_grid_dims', 'dst_grid_dims',
                              'row', 'col', 'S', 'frac_b', 'csr_indptr',
                              'csr_indices', 'csr_data')}


def reference_group_schedule(plan, grid_dims=None, super_tile=32, rows=8):
    """
    The row-group schedule of kernel family 10 written out with plain torch
    operations on the host's view of the CSR -- an independent restatement of
    what ``remap_groups_build`` (csrc/remap_schedule.hip) produces on the
    device: ``(meta, col, mask, w, rid, frac, order, n_union)``.
    """
    import torch
    G = int(rows)
    gx = G // 2
    dev = plan.device
    if grid_dims is not None and len(grid_dims) == 2:
        my, mx = (int(d) for d in grid_dims)
        st = int(super_tile) if super_tile and super_tile < 1 << 30 \
            else 1 << 30
        r = torch.arange(plan.row_offset, plan.row_offset + plan.n_b,
                         device=dev, dtype=torch.int64)
        jy = r // mx
        jx = r - jy * mx
        nsx = (mx + st - 1) // st
        key = ((jy // st) * nsx + jx // st) * (st * st) + \
            (((jy % st) // 2) * (st // gx) + (jx % st) // gx) * G + \
            (jy % 2) * gx + jx % gx
        order = torch.argsort(key, stable=True).to(torch.int32)
        slot_of_row = torch.empty(plan.n_b, dtype=torch.int64, device=dev)
        slot_of_row[order.to(torch.int64)] = torch.arange(plan.n_b,
                                                          device=dev)
    else:
        order = None
        slot_of_row = torch.arange(plan.n_b, device=dev)
    lens = plan.rowptr[1:] - plan.rowptr[:-1]
    entry_slot = torch.repeat_interleave(slot_of_row, lens)
    group_of_entry = entry_slot // G
    member = entry_slot % G
    n_groups = (plan.n_b + G - 1) // G
    key = group_of_entry * plan.n_a + plan.col.to(torch.int64)
    uniq, inverse = torch.unique(key, sorted=True, return_inverse=True)
    nu = int(uniq.shape[0])
    meta = torch.zeros((n_groups + 1, 2), dtype=torch.int64, device=dev)
    meta[1:, 0] = torch.cumsum(torch.bincount(uniq // plan.n_a,
                                              minlength=n_groups), 0)
    meta[1:, 1] = torch.cumsum(torch.bincount(group_of_entry,
                                              minlength=n_groups), 0)
    perm = torch.argsort(inverse * G + member)
    w = plan.val[perm]
    mask = torch.zeros(nu, dtype=torch.int32, device=dev)
    mask.index_add_(0, inverse, (1 << member).to(torch.int32))
    col = (uniq % plan.n_a).to(torch.int32)
    rid = torch.full((n_groups * G,), max(plan.n_b - 1, 0),
                     dtype=torch.int32, device=dev)
    rid[:plan.n_b] = order if order is not None else torch.arange(
        plan.n_b, device=dev, dtype=torch.int32)
    frac = plan.frac_b[rid.to(torch.int64)]
    return meta, col, mask, w, rid, frac, order, nu


def reference_patch_plan(plan, grid_dims, tile, rows_hint=None):
    """
    The LDS patch plan of kernel family 5 for a FIXED tile, written out with
    plain torch operations -- an independent restatement of what
    ``remap_patches_build`` (csrc/remap_schedule.hip) produces on the device:
    ``(ptr, ucol, rowptr, lidx, val, order, distinct, umax, emax)``.
    """
    import torch
    dev = plan.device
    ty, tx = (int(t) for t in tile)
    lens = plan.rowptr[1:] - plan.rowptr[:-1]
    entry_row = torch.repeat_interleave(
        torch.arange(plan.n_b, device=dev), lens)
    col64 = plan.col.to(torch.int64)
    if grid_dims is not None and len(grid_dims) == 2:
        my, mx = (int(d) for d in grid_dims)
        r = torch.arange(plan.row_offset, plan.row_offset + plan.n_b,
                         device=dev, dtype=torch.int64)
        jy = r // mx
        jx = r - jy * mx
        ntx = (mx + tx - 1) // tx
        key = ((jy // ty) * ntx + jx // tx) * (ty * tx) + \
            (jy % ty) * tx + jx % tx
        order = torch.argsort(key, stable=True).to(torch.int32)
        slot_of_row = torch.empty(plan.n_b, dtype=torch.int64, device=dev)
        slot_of_row[order.to(torch.int64)] = torch.arange(plan.n_b,
                                                          device=dev)
    else:
        order = None
        slot_of_row = torch.arange(plan.n_b, device=dev)
    rows = ty * tx
    n_patches = (plan.n_b + rows - 1) // rows
    patch_of_entry = slot_of_row[entry_row] // rows
    key = patch_of_entry * plan.n_a + col64
    uniq, inverse = torch.unique(key, sorted=True, return_inverse=True)
    counts = torch.bincount(uniq // plan.n_a, minlength=n_patches)
    ptr = torch.zeros(n_patches + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(counts, 0)
    lidx = (inverse - ptr[patch_of_entry]).to(torch.int32)
    rows_by_slot = order.to(torch.int64) if order is not None else \
        torch.arange(plan.n_b, device=dev)
    lens_by_slot = lens[rows_by_slot]
    prow = torch.zeros(plan.n_b + 1, dtype=torch.int64, device=dev)
    prow[1:] = torch.cumsum(lens_by_slot, 0)
    shift = plan.rowptr[:-1][rows_by_slot] - prow[:-1]
    src = torch.repeat_interleave(shift, lens_by_slot) + \
        torch.arange(plan.nnz, device=dev)
    per_patch = prow[torch.arange(0, n_patches * rows + 1, rows,
                                  device=dev).clamp(max=plan.n_b)]
    return (ptr.to(torch.int32), (uniq % plan.n_a).to(torch.int32),
            prow.to(torch.int32), lidx[src], plan.val[src], order,
            int(uniq.shape[0]), int(counts.max()),
            int((per_patch[1:] - per_patch[:-1]).max()))
