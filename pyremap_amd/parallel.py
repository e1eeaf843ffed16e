"""
Multi-GPU application of one mapping: destination rows sharded over GPUs.

The path partitions naturally -- destination row i needs only CSR row i and
the source rows it references (``remap_numpy.py:264-268``) -- so:

* every GPU holds the CSR rows of ONE contiguous, work-balanced range
  (:func:`row_shard_bounds`) and writes only its slab of Y;
* a shard's plan lives in the COMPACT space of the source rows it references
  (``RemapPlan.packed`` -> ``remap_pack_columns``): the shard's kernel reads
  the packed buffer ``X[ucols]`` (``engine.gather_rows`` ->
  ``remap_gather_rows``) and nothing else.  This holds for ANY numbering of
  the source mesh; the ``(min, max)`` band of source rows that round 2 sent
  instead only shrinks when the source cells are numbered along the
  destination raster, which no MPAS mesh does (one 1-degree row of the real
  QU240 mesh meets ids from 77-99 % of the id range: a band is all of X);
* the only exchange is the source field reaching the GPUs once per batch --
  each shard's packed rows (about ``(1/N + halo) |X|`` in all), or one
  broadcast of X followed by a local gather; no reduction collective;
* the slabs are gathered only where a single tensor is wanted.

Two front ends over the same shards:

* :class:`ShardedRemap` -- one process per GPU inside a ``torch.distributed``
  process group (backend ``nccl`` = RCCL over xGMI on MI355X nodes; ``gloo``
  for the CPU tests);
* :class:`MultiDeviceRemap` -- ONE process driving several GPUs (what a
  pyremap user has: ``remapper.remap_numpy(ds)`` from one Python process,
  ``remapper.py:508-532``); packed rows travel by peer copies.

The zero-collective alternative -- every rank remaps its own fields with
replicated weights -- needs no code here: each rank simply uses the unsharded
plan on its share of the fields.
"""
import logging
import os

log = logging.getLogger('pyremap_amd')


def _torch():
    import torch
    return torch


def _prod(seq):
    out = 1
    for s in seq:
        out *= int(s)
    return out


def row_shard_bounds(rowptr, world_size, row_cost=2):
    """
    ``world_size + 1`` row indices cutting ``[0, n_rows)`` into contiguous
    ranges of equal work, where work(range) = entries + ``row_cost`` * rows
    (the output write and first-touch source reads scale with rows, the
    gather with entries).  ``rowptr`` is an int64 tensor on any device.
    """
    torch = _torch()
    n_rows = int(rowptr.shape[0]) - 1
    if world_size <= 1 or n_rows <= 0:
        return [0] + [max(n_rows, 0)] * max(world_size, 1)
    rows = torch.arange(n_rows + 1, device=rowptr.device, dtype=torch.int64)
    work = rowptr.to(torch.int64) + row_cost * rows
    total = int(work[-1])
    targets = torch.tensor(
        [total * r // world_size for r in range(1, world_size)],
        device=rowptr.device, dtype=torch.int64)
    cuts = torch.searchsorted(work, targets).cpu().tolist()
    bounds = [0] + [min(int(c), n_rows) for c in cuts] + [n_rows]
    for i in range(1, len(bounds)):      # monotone even for degenerate input
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def broadcast_field(field, src=0, group=None):
    """
    The exchange in its collective form: ``src``'s source field goes to every
    rank (in place; ``field`` must be allocated with the same shape
    everywhere).
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and \
            dist.get_world_size(group) > 1:
        dist.broadcast(field, src=src, group=group)
    return field


def unique_columns(col, n_a=None):
    """
    Ascending distinct values of a shard's column indices (any device; plain
    torch -- the CPU tests' stand-in for ``RemapPlan.packed``).
    """
    torch = _torch()
    return torch.unique(col.to(torch.int64), sorted=True)


def packed_fraction(counts, n_a):
    """
    Source rows the packed exchange delivers, as a fraction of what one
    broadcast delivers (a whole field to every rank): ``sum(|ucols_r|) /
    (N * n_a)``.  About ``(1/N + halo)`` when neighbouring destination rows
    share source cells, whatever the numbering.
    """
    counts = list(counts)
    if not counts or n_a <= 0:
        return 1.0
    return sum(int(c) for c in counts) / (len(counts) * n_a)


def scatter_packed(pieces, recv_numel, dtype, device, src=0, group=None,
                   async_op=False):
    """
    ``src`` sends rank r the flat tensor ``pieces[r]`` (its packed source
    rows), every rank receives ``recv_numel`` elements: ONE
    ``all_to_all_single`` whose only non-empty sends leave ``src`` --
    ``alltoallv``, the collective RCCL runs for every expert-parallel model,
    rather than hand-rolled send / recv pairs.  On xGMI (point to point,
    7 links x ~153 GB/s per GPU) the pieces leave ``src`` over all links at
    once and sum to ``(1 + halo) |X|``; a broadcast ring moves all of X over
    every hop.

    ``pieces``: list of ``world`` flat tensors on ``src`` (ignored
    elsewhere).  Returns ``(recv, work)``; ``work`` is ``None`` unless
    ``async_op``.
    """
    torch = _torch()
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    recv = torch.empty(int(recv_numel), dtype=dtype, device=device)
    out_splits = [0] * world
    out_splits[src] = int(recv_numel)
    if rank == src:
        in_splits = [int(p.numel()) for p in pieces]
        send = torch.cat([p.reshape(-1) for p in pieces]) if world > 1 \
            else pieces[0].reshape(-1)
    else:
        in_splits = [0] * world
        send = torch.empty(0, dtype=dtype, device=device)
    work = dist.all_to_all_single(recv, send, out_splits, in_splits,
                                  group=group, async_op=async_op)
    return recv, (work if async_op else None)


def gather_rows(y_local, bounds, row_axis=0, group=None):
    """
    All ranks receive the full destination field: the slabs of rows
    ``[bounds[r], bounds[r + 1])`` concatenated along ``row_axis``.
    """
    torch = _torch()
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or \
            dist.get_world_size(group) == 1:
        return y_local
    world = dist.get_world_size(group)
    # all_gather wants equal shapes: pad every slab to the largest shard
    counts = [bounds[r + 1] - bounds[r] for r in range(world)]
    y_local = y_local.movedim(row_axis, 0).contiguous()
    padded_shape = [max(counts)] + list(y_local.shape[1:])
    mine = torch.zeros(padded_shape, dtype=y_local.dtype,
                       device=y_local.device)
    mine[:y_local.shape[0]] = y_local
    slabs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(slabs, mine, group=group)
    full = torch.cat([slab[:n] for slab, n in zip(slabs, counts)], dim=0)
    return full.movedim(0, row_axis)


def _flatten_source_axes(field, remap_axes):
    """
    View ``field`` as ``(n_batch, n_src, inner)`` with the source axes
    flattened into axis 1 -- in place when they are adjacent, through one
    permute copy to ``(1, n_src, K)`` otherwise (``remap_numpy.py:254-256``).
    Returns ``(x3, lead_shape, tail_shape, unpermute)``; ``unpermute`` is
    ``None`` or the function that puts a ``(1, n_dst..., K)``-shaped result
    back into the reference's axis order (``:280-295``).
    """
    ndim = field.ndim
    axes = [int(a) % ndim for a in remap_axes]
    lead = min(axes)
    n_src = _prod(field.shape[a] for a in axes)
    if axes == list(range(lead, lead + len(axes))):
        lead_shape = [int(s) for s in field.shape[:lead]]
        tail_shape = [int(s) for s in field.shape[lead + len(axes):]]
        x3 = field.contiguous().reshape(_prod(lead_shape), n_src,
                                        _prod(tail_shape))
        return x3, lead_shape, tail_shape, None
    extra = [a for a in range(ndim) if a not in axes]
    extra_shape = [int(field.shape[a]) for a in extra]
    x3 = field.permute(axes + extra).reshape(1, n_src, _prod(extra_shape)) \
        .contiguous()

    def unpermute(y, dst_shape):
        n_dst = len(dst_shape)
        tail = list(range(n_dst, n_dst + len(extra_shape)))
        order = tail[:lead] + list(range(n_dst)) + tail[lead:]
        return y.reshape(list(dst_shape) + extra_shape).permute(order) \
            .contiguous()
    return x3, [], extra_shape, unpermute


class _Shard:
    """Rows ``[r0, r1)`` of a mapping in their packed column space."""

    def __init__(self, full_plan, r0, r1, device=None, grid_dims=None):
        self.r0, self.r1 = int(r0), int(r1)
        sliced = full_plan.row_slice(self.r0, self.r1)
        packed, ucols = sliced.packed()
        #: distinct source rows this shard reads, ascending, on the FULL
        #: plan's device (where the field is gathered)
        self.ucols = ucols
        if device is not None and _torch().device(device) != packed.device:
            packed = packed.to(device)
        self.plan = packed
        self.schedule = packed.auto_schedule(grid_dims) \
            if grid_dims is not None else None


class ShardedRemap:
    """
    One rank's share of a row-sharded remap inside a process group.

    >>> sharded = ShardedRemap(full_plan, grid_dims=(360, 720))
    >>> xp = sharded.distribute(x, src=0)         # this rank's packed rows
    >>> y_rows = sharded.apply(xp, [0], mode)     # this rank's rows of Y
    >>> y = sharded.gather(y_rows)                # optional

    Every rank builds the full plan (the mapping file is small next to the
    fields) and keeps only its rows, in the compact space of the source rows
    they reference; with ``grid_dims`` (the destination grid of the WHOLE
    mapping) it picks the kernel schedule for them
    (``RemapPlan.auto_schedule``), as ``bench.py --gpus N`` does.

    ``exchange``: how :meth:`distribute` delivers the packed rows when asked
    for ``how='auto'`` -- ``'alltoall'`` (each rank receives only its packed
    rows: one ``all_to_all_single``) or ``'broadcast'`` (one broadcast of the
    whole field, then a local gather).  Default: the environment variable
    ``PYREMAP_AMD_EXCHANGE`` if set, else ``'broadcast'`` on the ``nccl``
    backend -- the all-to-all form has run under gloo and with ranks sharing
    one GPU only, not yet across xGMI -- and ``'alltoall'`` elsewhere.
    """

    def __init__(self, plan, group=None, grid_dims=None, rank=None,
                 world_size=None, exchange=None):
        import torch.distributed as dist
        self.group = group
        in_group = dist.is_available() and dist.is_initialized()
        if rank is not None and world_size is not None:
            # ranks managed by the caller (MPI launchers, tests)
            self.rank, self.world_size = int(rank), int(world_size)
        elif in_group:
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1
        self._full = plan
        self._grid_dims = grid_dims
        self.bounds = row_shard_bounds(plan.rowptr, self.world_size)
        shard = _Shard(plan, self.bounds[self.rank],
                       self.bounds[self.rank + 1], grid_dims=grid_dims)
        self.plan = shard.plan
        self.ucols = shard.ucols
        self.schedule = shard.schedule
        self._all_ucols = None
        self._counts = None
        backend = dist.get_backend(group) if in_group else None
        if exchange is None:
            exchange = os.environ.get('PYREMAP_AMD_EXCHANGE') or (
                'broadcast' if backend == 'nccl' else 'alltoall')
        if exchange not in ('alltoall', 'broadcast'):
            raise ValueError(f'unknown exchange {exchange!r}')
        self.exchange = exchange
        self._logged = set()

    # -- what each rank needs -------------------------------------------------
    def all_ucols(self):
        """Every rank's ``ucols`` (computed locally: each rank holds the
        full plan), as a list of int32 tensors on this rank's device."""
        if self._all_ucols is None:
            out = []
            for r in range(self.world_size):
                if r == self.rank:
                    out.append(self.ucols)
                else:
                    sl = self._full.row_slice(self.bounds[r],
                                              self.bounds[r + 1])
                    out.append(sl.packed()[1])
            self._all_ucols = out
        return self._all_ucols

    def packed_counts(self):
        """``len(ucols)`` of every rank."""
        if self._counts is None:
            self._counts = [int(u.shape[0]) for u in self.all_ucols()]
        return self._counts

    def packed_fraction(self):
        return packed_fraction(self.packed_counts(), self._full.n_a)

    # -- the exchange ---------------------------------------------------------
    def _packed_shape(self, shape, axis, n):
        shape = list(shape)
        shape[axis] = int(n)
        return shape

    def distribute(self, field, src=0, axis=0, how='auto', async_op=False,
                   shape=None, dtype=None):
        """
        The path's one exchange step: returns THIS rank's packed source rows
        ``field.index_select(axis, self.ucols)`` with ``field`` held by
        ``src`` only.

        ``how='alltoall'``: ``src`` gathers every rank's packed rows
        (``remap_gather_rows``) and one ``all_to_all_single`` delivers them;
        the other ranks pass ``field=None`` with ``shape`` / ``dtype`` (or a
        like-shaped tensor, which is not touched).  ``how='broadcast'``:
        ``field`` is a full-size buffer on every rank (in-place RCCL
        broadcast), each rank gathers its own rows.  ``how='auto'``:
        ``self.exchange``.  With ``async_op`` returns ``(packed, works)``:
        wait on ``works`` before the first launch that reads ``packed``.
        """
        from pyremap_amd import engine
        torch = _torch()
        import torch.distributed as dist
        if how == 'auto':
            how = self.exchange
        if how not in self._logged:
            self._logged.add(how)
            # (this rank's share only: `packed_fraction()` would pack every
            # other rank's shard here, which the broadcast form never needs)
            log.info(
                'ShardedRemap rank %d/%d: source field delivered by %s '
                '(this rank reads %.1f %% of the source rows)', self.rank,
                self.world_size, how, 100.0 * int(self.ucols.shape[0]) /
                max(self._full.n_a, 1))
        if field is not None:
            shape, dtype = tuple(field.shape), field.dtype
        axis = int(axis) % len(shape)
        if self.world_size == 1:
            packed = engine.gather_rows(field, axis, self.ucols)
            return (packed, []) if async_op else packed
        if how == 'broadcast':
            if field is None:
                field = torch.empty(shape, dtype=dtype,
                                    device=self.plan.device)
            work = dist.broadcast(field, src=src, group=self.group,
                                  async_op=async_op)
            if async_op:
                # the gather has to follow the broadcast in stream order:
                # done in wait_packed
                return _Deferred(field, axis, self.ucols), [work]
            return engine.gather_rows(field, axis, self.ucols)
        if how != 'alltoall':
            raise ValueError(f'unknown exchange {how!r}')
        pieces = None
        if self.rank == src:
            pieces = [engine.gather_rows(field, axis, u)
                      for u in self.all_ucols()]
        n_mine = int(self.ucols.shape[0])
        out_shape = self._packed_shape(shape, axis, n_mine)
        recv, work = scatter_packed(
            pieces, _prod(out_shape), dtype, self.plan.device, src=src,
            group=self.group, async_op=async_op)
        packed = recv.reshape(out_shape)
        return (packed, [work]) if async_op else packed

    def apply_pipelined(self, batches, mode, src=0, how='auto',
                        threshold=0.0, flags=0, outs=None):
        """
        Remap a sequence of field batches (each ``(n_a, K_b)``, rows on axis
        0; only ``src`` holds the data, the others pass like-shaped buffers)
        with the exchange of batch b + 1 in flight while batch b is computed:
        RCCL works on its own streams, the kernel on torch's current stream,
        and each launch waits only for its own batch's requests.  Returns the
        list of this rank's output slabs.
        """
        from pyremap_amd import engine
        results = []

        def start(x):
            return self.distribute(x, src=src, axis=0, how=how,
                                   async_op=True, shape=tuple(x.shape),
                                   dtype=x.dtype)

        pending = start(batches[0]) if batches else None
        for b, _ in enumerate(batches):
            packed, works = pending
            for work in works:
                work.wait()         # stream-orders the launch behind batch b
            if isinstance(packed, _Deferred):
                packed = packed.gather()
            pending = start(batches[b + 1]) if b + 1 < len(batches) else None
            results.append(engine.remap_tensor(
                self.plan, None, packed, [0], mode, threshold=threshold,
                flags=flags, out=None if outs is None else outs[b]))
        return results

    def apply(self, packed, remap_axes, mode, threshold=0.0, flags=0,
              tune=None, out=None, **kw):
        """This rank's rows of Y from ITS packed source rows."""
        from pyremap_amd import engine
        return engine.remap_tensor(self.plan, None, packed, remap_axes, mode,
                                   threshold=threshold, flags=flags,
                                   tune=tune, out=out, **kw)

    def gather(self, y_local, row_axis=0):
        return gather_rows(y_local, self.bounds, row_axis=row_axis,
                           group=self.group)

    # -- `_remap_numpy_array` as a collective ----------------------------------
    def remap_tensor(self, dst_grid_dims, field, remap_axes, threshold=None,
                     src=0, flags=0, shape=None, dtype=None, mode='auto',
                     want_mask=False):
        """
        The whole array-level remap as a COLLECTIVE call: every rank calls
        it with the same arguments, ``src`` passes the device tensor (the
        others ``None`` plus ``shape`` / ``dtype``, or any tensor of that
        shape), every rank returns the full float64 result (NaN where the
        reference masks).  ``mode='auto'``: ``threshold`` ``None`` = the
        unmasked branch, else masked iff the field holds a NaN
        (``remap_numpy.py:201-204``); ``'masked'`` / ``'fracb'`` / ``'raw'``:
        that branch.  With ``want_mask`` returns ``(y, mask)``, ``mask`` the
        reference's output mask (``:278``: ``den <= threshold`` in masked
        mode, ``frac_b <= 0`` otherwise) as uint8, gathered like ``y``.
        """
        from pyremap_amd import engine
        torch = _torch()
        import torch.distributed as dist
        if field is not None:
            shape, dtype = tuple(field.shape), field.dtype
        like = field if field is not None and self.rank == src else \
            torch.empty(shape, dtype=dtype, device='meta')
        # (a rank that only knows the shape goes through the same view
        # arithmetic on a meta tensor)
        x3, lead_shape, tail_shape, unpermute = _flatten_source_axes(
            like, remap_axes)
        hint = 0
        if mode == 'auto':
            masked = False
            if threshold is not None:
                # the branch (remap_numpy.py:201-204) and the form of the
                # masked launch that suits the field -- whole cells missing,
                # the same mask in every batch -- from ONE layout-aware scan
                # on `src`, four words to every rank
                kinds = torch.zeros(4, dtype=torch.int32,
                                    device=self.plan.device)
                if self.rank == src:
                    x3c = x3.contiguous()
                    if x3c.dtype not in (torch.float64, torch.float32):
                        x3c = x3c.to(torch.float64)
                    engine.scan_nan_layout(x3c, int(x3c.shape[1]),
                                           int(x3c.shape[0]),
                                           int(x3c.shape[2]), kinds)
                if self.world_size > 1:
                    dist.broadcast(kinds, src=src, group=self.group)
                found = kinds.tolist()
                masked = bool(found[0])
                hint = {1: engine.FLAG_CELL_MASKS,
                        2: engine.FLAG_BATCH_MASKS}.get(found[3], 0)
            emode = engine.MODE_MASKED if masked else engine.MODE_FRACB
        else:
            emode = {'masked': engine.MODE_MASKED, 'fracb': engine.MODE_FRACB,
                     'raw': engine.MODE_RAW}[mode]
        packed = self.distribute(
            x3 if self.rank == src else None, src=src, axis=1,
            shape=tuple(x3.shape), dtype=dtype)
        res = engine.remap_tensor(
            self.plan, None, packed, [1], emode,
            threshold=float(threshold) if emode == engine.MODE_MASKED
            else 0.0, flags=(flags | hint) if emode == engine.MODE_MASKED
            else flags, want_mask=want_mask)
        dst_shape = [int(d) for d in dst_grid_dims] \
            if dst_grid_dims is not None else [self._full.n_b]

        def whole(part):
            part = self.gather(part, row_axis=1)
            if unpermute is not None:
                return unpermute(part, dst_shape)
            return part.reshape(lead_shape + dst_shape + tail_shape)
        if want_mask:
            return whole(res[0]), whole(res[1])
        return whole(res)


class _Deferred:
    """Packed rows to be gathered once an in-flight broadcast has landed."""

    def __init__(self, field, axis, ucols):
        self.field, self.axis, self.ucols = field, axis, ucols

    def gather(self):
        from pyremap_amd import engine
        return engine.gather_rows(self.field, self.axis, self.ucols)


class MultiDeviceRemap:
    """
    ONE process, several GPUs: the ``Remapper(..., devices=[...])`` mode.

    The destination rows of ``plan`` (a full ``RemapPlan`` on
    ``plan.device``, the *source device*: where fields are handed over and
    results are returned) are cut into one work-balanced range per entry of
    ``devices``; each shard lives on its device in its packed column space
    with its own schedule.  Per call the source device gathers each shard's
    packed rows (``remap_gather_rows``), peer copies carry them over xGMI,
    every device launches on its own stream, and the slabs come back into
    one tensor on the source device (``gather=True``) or stay where they are.
    The same device may be listed several times (tests emulate N devices on
    one GPU that way).

    It stands where the single-device plan stands on a Remapper
    (``remapper._matrix``): ``n_a``, ``n_b``, ``n_b_global``, ``device``,
    :meth:`remap_tensor`, :meth:`remap_tensor_auto_mode`.
    """

    def __init__(self, plan, devices, grid_dims=None):
        torch = _torch()
        self.devices = [torch.device(d) for d in devices]
        if not self.devices:
            raise ValueError('devices must name at least one GPU')
        for d in self.devices:
            if d.type != 'cuda':
                raise ValueError(f'{d} is not a GPU')
        self.device = plan.device
        self.n_a, self.n_b = plan.n_a, plan.n_b
        self.n_b_global = plan.n_b_global
        self.nnz = plan.nnz
        self.bounds = row_shard_bounds(plan.rowptr, len(self.devices))
        self.shards = [
            _Shard(plan, self.bounds[i], self.bounds[i + 1], device=d,
                   grid_dims=grid_dims)
            for i, d in enumerate(self.devices)]
        self.schedule = [s.schedule for s in self.shards]
        log.info('MultiDeviceRemap: %d shards on %s, packed rows = %.1f %% '
                 'of a broadcast', len(self.shards),
                 [str(d) for d in self.devices],
                 100.0 * self.packed_fraction())

    def packed_fraction(self):
        return packed_fraction([s.ucols.shape[0] for s in self.shards],
                               self.n_a)

    def remap_tensor(self, dst_grid_dims, field, remap_axes, mode,
                     threshold=0.0, want_mask=False, flags=0, gather=True,
                     _gate=None):
        """
        ``engine.remap_tensor`` over the shards.  ``field``: a device tensor
        (moved to the source device if it lives elsewhere).  Returns the
        float64 result on the source device in the reference's axis order
        (``remap_numpy.py:280-295``) -- and the uint8 mask with
        ``want_mask`` -- or, with ``gather=False``, the list of per-device
        slabs ``(n_batch, rows_d, inner)`` in shard order.
        """
        from pyremap_amd import engine
        torch = _torch()
        field = field.to(self.device)
        if field.dtype not in (torch.float64, torch.float32):
            field = field.to(torch.float64)
        x3, lead_shape, tail_shape, unpermute = _flatten_source_axes(
            field, remap_axes)
        if x3.shape[1] != self.n_a:
            raise ValueError(
                f'the remapped axes hold {x3.shape[1]} source cells but the '
                f'mapping has n_a = {self.n_a}')
        n_batch, inner = int(x3.shape[0]), int(x3.shape[2])
        if _gate is not None:
            # remap_numpy.py:201-204 decided on the device, ONCE for the
            # whole field (the branch is a property of the field, not of a
            # shard's packed rows): the layout-aware scan on the source
            # device -- any NaN; whole cells missing; the same mask in every
            # batch; the masked form that suits (engine.scan_nan_layout)
            with torch.cuda.device(self.device):
                x3 = x3.contiguous()
                _gate = torch.zeros(4, dtype=torch.int32, device=self.device)
                engine.scan_nan_layout(x3, self.n_a, n_batch, inner, _gate)
        slabs, masks = [], []
        for shard in self.shards:
            dev = shard.plan.device
            with torch.cuda.device(self.device):
                xp = engine.gather_rows(x3, 1, shard.ucols)
            xp = xp.to(dev, non_blocking=True)
            rows = shard.r1 - shard.r0
            with torch.cuda.device(dev):
                y = torch.empty((n_batch, rows, inner), dtype=torch.float64,
                                device=dev)
                m = torch.empty((n_batch, rows, inner), dtype=torch.uint8,
                                device=dev) if want_mask else None
                if _gate is None:
                    engine.remap_tensor(
                        shard.plan, None, xp, [1], mode, threshold=threshold,
                        want_mask=want_mask, flags=flags, out=y, mask_out=m)
                else:
                    # every candidate launch enqueued, each gated on the
                    # scan's words (copied to the shard's device): frac_b,
                    # and the masked mode in the form that suits -- as
                    # engine.remap_tensor_auto_mode does on one device
                    gate = _gate.to(dev, non_blocking=True)
                    hints = engine.FLAG_CELL_MASKS | engine.FLAG_BATCH_MASKS
                    base = flags & ~hints
                    if engine.cell_mask_form(shard.plan):
                        forms = [(engine.FLAG_CELL_MASKS, 1)]
                        if n_batch >= 3:
                            forms.append((engine.FLAG_BATCH_MASKS, 2))
                        forms.append((0, 3))
                        for hint, value in forms:
                            engine.remap_tensor(
                                shard.plan, None, xp, [1],
                                engine.MODE_MASKED, threshold=threshold,
                                flags=base | hint, out=y, gate=gate[3:],
                                gate_value=value)
                    else:
                        engine.remap_tensor(
                            shard.plan, None, xp, [1], engine.MODE_MASKED,
                            threshold=threshold, flags=base, out=y,
                            gate=gate, gate_value=1)
                    engine.remap_tensor(
                        shard.plan, None, xp, [1], engine.MODE_FRACB,
                        flags=base, out=y, gate=gate, gate_value=0)
            slabs.append(y)
            masks.append(m)
        if not gather:
            return (slabs, masks) if want_mask else slabs
        dst_shape = [int(d) for d in dst_grid_dims] \
            if dst_grid_dims is not None else [self.n_b]

        def assemble(parts, dtype):
            with torch.cuda.device(self.device):
                full = torch.empty((n_batch, self.n_b, inner), dtype=dtype,
                                   device=self.device)
                for shard, part in zip(self.shards, parts):
                    full[:, shard.r0:shard.r1].copy_(part, non_blocking=True)
            if unpermute is not None:
                return unpermute(full, dst_shape)
            return full.reshape(lead_shape + dst_shape + tail_shape)

        y = assemble(slabs, torch.float64)
        return (y, assemble(masks, torch.uint8)) if want_mask else y

    def remap_tensor_auto_mode(self, dst_grid_dims, field, remap_axes,
                               threshold, flags=0):
        """``engine.remap_tensor_auto_mode`` over the shards: one NaN scan
        of the whole field on the source device (layout-aware: it also names
        the form of the masked launch), the gated launches per shard -- two,
        or four on entry-rich mappings."""
        from pyremap_amd import engine
        return self.remap_tensor(dst_grid_dims, field, remap_axes,
                                 engine.MODE_MASKED, threshold=threshold,
                                 flags=flags, _gate=True)
