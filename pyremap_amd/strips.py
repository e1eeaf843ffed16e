"""
The strip schedule of kernel family 8 (``csrc/spmm_strip.h``,
``include/remap_hip.h: remap_strips``): for entry-rich mappings onto a 2-D
destination grid -- 2nd-order conservative stencils, 12-30 entries per row
(BASELINE config 5) -- where the wave-per-row(-group) kernels are bound by
L2 -> L1 fills (every group of rows pulls its stencil through the CU again).

The destination grid is cut into strips of ``strip_rows`` grid rows, a strip
into ``segments``; one workgroup walks one (strip, segment) = *unit* in steps
of ``step_cols`` grid columns for one 64-column K-chunk.  The 512-byte pieces
of the source rows a step needs *arrive* in LDS slots ``depth`` steps ahead
and stay while the following steps use them, so what neighbouring steps share
is fetched once; only the halo above and below the strip is fetched by two
strips.  Slots are handed out in PAIRS (one LDS-DMA instruction carries two
pieces) from a free list: a pair is reused as soon as both its pieces are
dead, so the LDS holds little more than the pieces that are alive.

Built with torch on the plan's device (one-off per mapping, like
``RemapPlan.build_groups``); the arithmetic the kernel does with it is the
reference's: every destination row adds its entries in ascending column
order (``remap_numpy.py:264-268``, scipy's ``csr_matvecs``).  Rows are padded
to a multiple of ``QUANT`` entries with records of weight +0.0 that point at
a slot of zeros: ``acc + (+0.0 * +0.0)`` leaves every ``acc`` this sum can
hold unchanged bit for bit (the running sum starts at +0.0 and therefore
never is -0.0), so the kernel's inner loop needs no per-entry test.
"""
import ctypes

PIECE_BYTES = 512    # bytes of a source row in LDS: 64 float64 columns
LDS_BYTES = 160 * 1024
QUANT = 4            # rows hold a multiple of QUANT records (spmm_strip.h)


class StripsUnfit(ValueError):
    """The mapping does not fit a strip schedule of the asked shape."""


def _snake(n, waves):
    """position in the (descending-length) order -> (wave, index in wave),
    dealt back and forth so that every wave gets long and short rows."""
    import torch
    pos = torch.arange(n)
    rnd = pos // waves
    w = pos % waves
    w = torch.where(rnd % 2 == 1, waves - 1 - w, w)
    return w, rnd


def _allocate_pairs(p_unit, p_step, p_expiry, n_units, spu, depth, n_pairs):
    """
    LDS slot pairs for the arrival pairs (``p_unit``, ``p_step``: when they
    are needed first; ``p_expiry``: the last step that reads one of their
    pieces), sorted by (step, unit).  The arrivals of step s are issued while
    step s - depth is computed: a pair is free for them once its pieces'
    last reader is a step before that.  Returns the pair index of every
    arrival pair, or None when ``n_pairs`` do not suffice.
    """
    import torch
    dev = p_unit.device
    never = -(1 << 30)
    expiry = torch.full((n_units, n_pairs), never, dtype=torch.int64,
                        device=dev)
    out = torch.empty(p_unit.shape[0], dtype=torch.int64, device=dev)
    bounds = torch.searchsorted(
        p_step, torch.arange(spu + 1, device=dev, dtype=p_step.dtype))
    bounds = bounds.tolist()
    cols = torch.arange(n_pairs, device=dev)
    for t in range(spu):
        lo, hi = bounds[t], bounds[t + 1]
        if lo == hi:
            continue
        u = p_unit[lo:hi]
        # index of each arrival pair among its unit's pairs of this step
        first = torch.searchsorted(u, u, right=False)
        q = torch.arange(hi - lo, device=dev) - first
        free = expiry < t - depth
        rank = torch.cumsum(free, 1) - 1              # rank of a free pair
        pos = torch.full((n_units, n_pairs + 1), -1, dtype=torch.int64,
                         device=dev)
        pos.scatter_(1, torch.where(free, rank, n_pairs), cols.expand_as(
            rank))
        if int(q.max()) >= n_pairs:
            return None
        got = pos[u, q]
        if bool((got < 0).any()):
            return None
        expiry[u, got] = p_expiry[lo:hi]
        out[lo:hi] = got
    return out


def build_strips(plan, grid_dims, strip_rows=14, step_cols=1, segments=4,
                 depth=2, gap=1, waves=14, max_bytes=None):
    """
    The strip schedule of ``plan`` (a whole mapping: ``row_offset == 0``) on
    the destination grid ``grid_dims = (my, mx)``.  Returns a dict of device
    tensors and scalars (``struct()`` of it: the ctypes ``remap_strips``).

    ``waves``: compute waves per workgroup (``waves + depth <= 16``); the
    rows of a step are dealt to them.  ``gap``: a piece stays over up to
    that many steps that do not read it (a halo cell is wanted by few rows
    of the strip, not by every step) instead of being fetched again.
    Raises :class:`StripsUnfit` when the pieces alive at one time, the slot
    of zeros and the meta slots do not fit ``max_bytes`` of LDS.
    """
    import torch
    my, mx = (int(d) for d in grid_dims)
    n_b, n_a = plan.n_b, plan.n_a
    if my * mx != n_b:
        raise ValueError(f'grid {my} x {mx} != {n_b} destination rows')
    R, W, NW = int(strip_rows), int(step_cols), int(waves)
    if not 1 <= depth <= 6:
        raise ValueError('depth must be 1 ... 6')
    if NW < 1 or NW + depth > 16:
        raise ValueError('waves + depth must be at most 16')
    dev = plan.device
    rpw = -(-(R * W) // NW)                       # rows per wave and step
    rps = NW * rpw                                # row slots per step
    seg_cols = -(-mx // (segments * W)) * W       # columns per segment
    n_seg = -(-mx // seg_cols)
    n_strip = -(-my // R)
    spu = seg_cols // W                           # steps per unit (at most)
    n_units = n_strip * n_seg
    n_steps = n_units * spu

    rowptr = plan.rowptr.to(torch.int64)
    nnz = int(rowptr[-1])
    lens = (rowptr[1:] - rowptr[:-1])
    plen = (lens + QUANT - 1) // QUANT * QUANT    # padded record counts
    rows = torch.arange(n_b, device=dev, dtype=torch.int64)
    jy, jx = rows // mx, rows % mx
    unit = (jy // R) * n_seg + jx // seg_cols
    lstep = (jx % seg_cols) // W
    gstep = unit * spu + lstep                    # dense step id of a row
    # steps a unit really has (the last segment may be narrower)
    unit_cols = torch.clamp(
        mx - (torch.arange(n_units, device=dev) % n_seg) * seg_cols,
        max=seg_cols)
    unit_steps = (-(-unit_cols // W)).to(torch.int32)

    # ---- rows of a step: longest first, dealt to the waves back and forth
    order = torch.argsort(gstep * (1 << 20) + ((1 << 20) - 1 - lens),
                          stable=True)
    first = torch.searchsorted(gstep[order], gstep[order], right=False)
    pos = torch.arange(n_b, device=dev) - first
    sw, si = _snake(R * W, NW)
    sw, si = sw.to(dev), si.to(dev)
    work = (gstep[order] * NW + sw[pos]) * rpw + si[pos]
    n_work = n_steps * rps
    row_rid = torch.full((n_work,), -1, dtype=torch.int32, device=dev)
    row_rid[work] = order.to(torch.int32)
    row_len = torch.zeros(n_work, dtype=torch.int64, device=dev)
    row_len[work] = plen[order]
    row_fb = torch.zeros(n_work, dtype=torch.float64, device=dev)
    row_fb[work] = plan.frac_b[order]
    # records laid out in work-slot order (a wave's rows are contiguous)
    row_ent = torch.zeros(n_work + 1, dtype=torch.int64, device=dev)
    torch.cumsum(row_len, 0, out=row_ent[1:])
    del first, pos

    # ---- which (step, source row) pairs are needed, which of them ARRIVE
    erow = torch.repeat_interleave(rows, lens)    # destination row per entry
    ecol = plan.col[:nnz].to(torch.int64)
    # sorted by (unit, source row, step): a stay = a run of steps no more
    # than gap + 1 apart
    pkey = (unit[erow] * n_a + ecol) * spu + lstep[erow]
    ukey = torch.unique(pkey)                     # sorted
    n_need = ukey.shape[0]
    ustep = ukey % spu
    uuc = ukey // spu                             # unit * n_a + source row
    uunit = uuc // n_a
    ucol = uuc % n_a
    cont = torch.zeros(n_need, dtype=torch.bool, device=dev)
    if n_need > 1:
        cont[1:] = (uuc[1:] == uuc[:-1]) & ((ustep[1:] - ustep[:-1]) <=
                                            gap + 1)
    arrives = ~cont
    idx = torch.arange(n_need, device=dev)
    opener = torch.cummax(torch.where(arrives, idx, -1), 0).values
    # last step of every stay: the step of the need before the next opener
    a_idx = arrives.nonzero().squeeze(1)
    stay_end = torch.empty_like(a_idx)
    stay_end[:-1] = a_idx[1:] - 1
    stay_end[-1:] = n_need - 1
    a_last = ustep[stay_end]                      # local step
    a_unit, a_step, a_col = uunit[a_idx], ustep[a_idx], ucol[a_idx]
    # arrivals of one (step, unit) are paired: similar life times together
    a_order = torch.argsort((a_step * n_units + a_unit) * (spu + 1) + a_last)
    a_idx, a_unit, a_step, a_col, a_last = (
        v[a_order] for v in (a_idx, a_unit, a_step, a_col, a_last))
    a_gstep = a_unit * spu + a_step
    n_arr_real = a_idx.shape[0]
    su_key = a_step * n_units + a_unit
    grp_first = torch.searchsorted(su_key, su_key, right=False)
    within = torch.arange(n_arr_real, device=dev) - grp_first
    # pair table, sorted by (step, unit): pair id = running count of pairs
    is_pair_head = within % 2 == 0
    pair_of = torch.cumsum(is_pair_head, 0) - 1
    n_pairs_total = int(pair_of[-1]) + 1 if n_arr_real else 0
    p_unit = a_unit[is_pair_head]
    p_step = a_step[is_pair_head]
    p_expiry = torch.zeros(n_pairs_total, dtype=torch.int64, device=dev)
    p_expiry.scatter_reduce_(0, pair_of, a_last, 'amax', include_self=True)

    # ---- meta blocks: per step, rps headers of 32 bytes, then one 16-byte
    # record per (padded) entry in work-slot order
    step_ent = row_ent[::rps]                     # first record of each step
    units16 = torch.arange(n_steps + 1, device=dev, dtype=torch.int64) * \
        (rps * 2) + step_ent
    block16 = units16[1:] - units16[:-1]
    slot_bytes = int(-(-(int(block16.max()) * 16) // 1024) * 1024) \
        if n_steps else 1024
    limit = LDS_BYTES if max_bytes is None else max_bytes
    fixed = PIECE_BYTES * 2 + (depth + 1) * slot_bytes   # zeros + meta slots
    most = (limit - fixed) // (2 * PIECE_BYTES)
    # the fewest slot pairs that hold every piece while it is alive: start
    # from a lower bound, grow until the free list never runs dry
    per_step = torch.bincount(p_step * n_units + p_unit,
                              minlength=spu * n_units)
    if per_step.numel() and int(per_step.max()) > 64:
        raise StripsUnfit(f'{int(per_step.max())} arrival pairs in one step '
                          f'(the loader holds a list of 64)')
    n_slot_pairs = max(int(per_step.max()) if per_step.numel() else 1, 4)
    pair_slot = None
    while n_slot_pairs <= most:
        pair_slot = _allocate_pairs(p_unit, p_step, p_expiry, n_units, spu,
                                    depth, n_slot_pairs)
        if pair_slot is not None:
            break
        n_slot_pairs = n_slot_pairs + max(2, n_slot_pairs // 8)
    if pair_slot is None:
        raise StripsUnfit(
            f'more than {most} slot pairs ({limit} bytes of LDS less '
            f'{fixed} for the meta slots) needed with strips of {R} rows, '
            f'steps of {W} columns and depth {depth}')
    cap = 2 * n_slot_pairs
    total = cap * PIECE_BYTES + fixed
    # the loaders' lists: per step, pairs of source rows and their slot pair
    pg = p_unit * spu + p_step                    # dense step of each pair
    g_order = torch.argsort(pg, stable=True)
    counts = torch.bincount(pg, minlength=n_steps)
    arr_ptr = torch.zeros(n_steps + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=arr_ptr[1:])
    new_id = torch.empty(n_pairs_total, dtype=torch.int64, device=dev)
    new_id[g_order] = torch.arange(n_pairs_total, device=dev)
    arr_slot = torch.zeros(n_pairs_total + 2, dtype=torch.int32, device=dev)
    arr_slot[new_id] = pair_slot.to(torch.int32)
    arr_src = torch.full((2 * n_pairs_total + 4,), -1, dtype=torch.int32,
                         device=dev)
    arr_src[new_id[pair_of] * 2 + within % 2] = a_col.to(torch.int32)
    # LDS byte offset of every needed (step, source row): its stay's slot
    a_off = (pair_slot[pair_of] * 2 + within % 2) * PIECE_BYTES
    need_off = torch.empty(n_need, dtype=torch.int64, device=dev)
    need_off[a_idx] = a_off
    need_off = need_off[opener]

    n16 = int(units16[-1])
    zero_off = cap * PIECE_BYTES                  # the slot of zeros
    meta = torch.zeros((n16 + 64, 2), dtype=torch.int64, device=dev)
    wslot = torch.arange(n_work, device=dev, dtype=torch.int64)
    wstep = wslot // rps
    hdr = units16[wstep] + (wslot % rps) * 2
    e_rel = row_ent[:-1] - step_ent[wstep]
    meta[hdr, 0] = (row_rid.to(torch.int64) & 0xffffffff) | (e_rel << 32)
    meta[hdr, 1] = row_len
    meta[hdr + 1, 0] = row_fb.view(torch.int64)
    # every record slot starts as a pad (weight +0.0, the slot of zeros) ...
    rec_base = units16[wstep] + rps * 2 + e_rel
    pad_row = torch.repeat_interleave(wslot, row_len)
    pad_pos = rec_base[pad_row] + (torch.arange(pad_row.shape[0], device=dev)
                                   - row_ent[:-1][pad_row])
    meta[pad_pos, 0] = zero_off
    del pad_row, pad_pos
    # ... and the entries overwrite the first `len` of their row's
    e_need = torch.searchsorted(ukey, pkey)
    work_of_row = torch.empty(n_b, dtype=torch.int64, device=dev)
    work_of_row[order] = work
    within_row = torch.arange(nnz, device=dev) - rowptr[:-1][erow]
    dest = rec_base[work_of_row[erow]] + within_row
    meta[dest, 0] = need_off[e_need]
    meta[dest, 1] = plan.val[:nnz].view(torch.int64)
    return dict(
        n_units=n_units, steps_per_unit=spu, rows_per_wave=rpw,
        ring_slots=cap, depth=int(depth), meta_slot_bytes=slot_bytes,
        waves=NW, strip_rows=R, step_cols=W, segments=n_seg, gap=int(gap),
        unit_steps=unit_steps, arr_ptr=arr_ptr.to(torch.int32),
        arr_src=arr_src, arr_slot=arr_slot, meta_ptr=units16, meta=meta,
        arrivals=n_arr_real, n_a=n_a, lds_bytes=total,
        records=int(row_ent[-1]))


def struct(strips, struct_type):
    """The ctypes ``remap_strips`` of a schedule (the dict keeps the tensors
    alive)."""
    st = struct_type()
    for name in ('n_units', 'steps_per_unit', 'rows_per_wave', 'ring_slots',
                 'depth', 'meta_slot_bytes', 'waves'):
        setattr(st, name, int(strips[name]))
    for name in ('unit_steps', 'arr_ptr', 'arr_src', 'arr_slot', 'meta_ptr',
                 'meta'):
        setattr(st, name, ctypes.c_void_p(strips[name].data_ptr()))
    return st
