"""
Multi-GPU application of one mapping: destination rows sharded over the
ranks of a ``torch.distributed`` process group (one process per GPU, backend
``nccl`` = RCCL over xGMI on MI355X nodes; ``gloo`` on CPU for tests).

The path partitions naturally -- destination row i needs only CSR row i and
the source rows it references (``remap_numpy.py:264-268``) -- so:

* every rank holds the CSR rows of ONE contiguous, work-balanced range
  (:func:`row_shard_bounds`) and writes only its slab of Y;
* the only exchange is the source field reaching the ranks once per batch:
  ONE broadcast (:func:`broadcast_field`) or, cheaper on point-to-point xGMI,
  each rank receiving only the band of source rows its shard references
  (:func:`distribute_rows`); successive batches are pipelined behind the
  kernel (:meth:`ShardedRemap.apply_pipelined`); no reduction collective;
* :func:`gather_rows` assembles the slabs where a single tensor is wanted
  (reported separately from the compute phase).

The zero-collective alternative -- every rank remaps its own fields with
replicated weights -- needs no code here: each rank simply uses the unsharded
plan on its share of the fields.
"""


def _torch():
    import torch
    return torch


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
    The path's one exchange step: ``src``'s source field goes to every rank
    (in place; ``field`` must be allocated with the same shape everywhere).
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and \
            dist.get_world_size(group) > 1:
        dist.broadcast(field, src=src, group=group)
    return field


def source_row_range(col, n_a=None):
    """
    ``(lo, hi)``: the half-open range of source rows a shard's entries
    reference (``col`` = the shard's column indices, any device).  On
    mappings whose destination order follows the source mesh -- every
    regridding map does, to the extent both grids cover the same sphere in a
    similar order -- a contiguous destination-row shard needs a contiguous
    band of about ``n_a / world`` source rows plus a halo, not all of X.
    """
    if col.numel() == 0:
        return 0, 0
    return int(col.min()), int(col.max()) + 1


def exchange_row_ranges(lo, hi, device=None, group=None):
    """Every rank's ``(lo, hi)`` (one small all_gather, at set-up time)."""
    torch = _torch()
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or \
            dist.get_world_size(group) == 1:
        return [(int(lo), int(hi))]
    world = dist.get_world_size(group)
    mine = torch.tensor([int(lo), int(hi)], dtype=torch.int64, device=device)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    return [tuple(int(v) for v in t.cpu()) for t in got]


def distribute_rows(field, ranges, src=0, group=None, async_op=False):
    """
    The exchange step without a broadcast: ``src`` sends every other rank
    ONLY the band of source rows its shard references (``ranges`` from
    :func:`exchange_row_ranges`; rows are axis 0 of ``field``, which every
    rank allocates at full size -- rows outside a rank's band are never read
    by its kernel and stay as they are).

    Why: xGMI is point to point (7 links x ~153 GB/s per GPU).  A broadcast
    ring moves the whole field over every hop, so it costs ``bytes(X) / one
    link``; the bands leave ``src`` over seven links at once and sum to about
    ``bytes(X) * (1 + halo)``, i.e. ~``bytes(X) / 7`` per link.

    Returns the list of outstanding requests when ``async_op`` (wait on them
    before the first launch that reads ``field``), else ``None``.
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or \
            dist.get_world_size(group) == 1:
        return [] if async_op else None
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    ops = []
    if rank == src:
        for r in range(world):
            lo, hi = ranges[r]
            if r != src and hi > lo:
                ops.append(dist.P2POp(dist.isend, field[lo:hi], r,
                                      group=group))
    else:
        lo, hi = ranges[rank]
        if hi > lo:
            ops.append(dist.P2POp(dist.irecv, field[lo:hi], src,
                                  group=group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return reqs
    for req in reqs:
        req.wait()
    return None


def band_fraction(ranges, n_a):
    """Bytes the bands move, as a fraction of one whole field per rank."""
    if not ranges or n_a <= 0:
        return 1.0
    return sum(hi - lo for lo, hi in ranges) / (len(ranges) * n_a)


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


class ShardedRemap:
    """
    One rank's share of a row-sharded remap.

    >>> sharded = ShardedRemap(full_plan, grid_dims=(360, 720))
    >>> x = sharded.broadcast(x)                  # inside an initialised
    >>> y_rows = sharded.apply(x, [0], mode)      # process group: this
    >>> y = sharded.gather(y_rows)                # rank's rows; optional

    With ``grid_dims`` (the destination grid of the WHOLE mapping) every rank
    picks the kernel schedule for its own rows (``RemapPlan.auto_schedule``),
    as ``bench.py --gpus N`` does.
    """

    def __init__(self, plan, group=None, grid_dims=None, rank=None,
                 world_size=None):
        import torch.distributed as dist
        self.group = group
        if rank is not None and world_size is not None:
            # ranks managed by the caller (MPI launchers, tests)
            self.rank, self.world_size = int(rank), int(world_size)
        elif dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1
        self.bounds = row_shard_bounds(plan.rowptr, self.world_size)
        self.plan = plan.row_slice(self.bounds[self.rank],
                                   self.bounds[self.rank + 1]) \
            if self.world_size > 1 else plan
        self.schedule = self.plan.auto_schedule(grid_dims) \
            if grid_dims is not None else None
        #: band of source rows this rank's rows reference, and everyone's
        self.src_range = source_row_range(self.plan.col)
        in_group = dist.is_available() and dist.is_initialized() and \
            dist.get_world_size(group) == self.world_size
        if self.world_size == 1:
            self.src_ranges = [self.src_range]
        elif in_group:
            self.src_ranges = exchange_row_ranges(
                *self.src_range, device=self.plan.device, group=group)
        else:
            self.src_ranges = None   # ranks managed by the caller: no group

    def broadcast(self, field, src=0):
        return broadcast_field(field, src=src, group=self.group)

    def distribute(self, field, src=0, how='auto', async_op=False):
        """
        The path's one exchange step.  ``how``: ``'broadcast'`` (the whole
        field to every rank, one RCCL broadcast), ``'bands'`` (each rank gets
        only the source rows its shard references, point to point) or
        ``'auto'``: bands when they move less than 60 % of what a broadcast
        delivers (on a raster-ordered mapping over 8 ranks: ~15 %).
        ``field``: rows on axis 0, allocated at full size on every rank.
        """
        if self.world_size == 1:
            return [] if async_op else field
        if self.src_ranges is None:
            raise RuntimeError('distribute() needs an initialised process '
                               'group spanning the ranks of this remap')
        if how == 'auto':
            how = 'bands' if band_fraction(
                self.src_ranges, self.plan.n_a) < 0.6 else 'broadcast'
        if how == 'bands':
            reqs = distribute_rows(field, self.src_ranges, src=src,
                                   group=self.group, async_op=async_op)
            return reqs if async_op else field
        import torch.distributed as dist
        work = dist.broadcast(field, src=src, group=self.group,
                              async_op=async_op)
        return [work] if async_op else field

    def apply_pipelined(self, batches, mode, src=0, how='auto',
                        threshold=0.0, flags=0, outs=None):
        """
        Remap a sequence of field batches (each ``(n_a, K_b)``, rows on axis
        0, allocated on every rank; only ``src`` holds the data) with the
        exchange of batch b + 1 in flight while batch b is computed: RCCL
        works on its own streams, the kernel on torch's current stream, and
        each launch waits only for its own batch's requests.  Returns the
        list of this rank's output slabs.
        """
        from pyremap_amd import engine
        results = []
        pending = self.distribute(batches[0], src=src, how=how,
                                  async_op=True) if batches else []
        for b, x in enumerate(batches):
            for req in pending:
                req.wait()          # stream-orders the launch behind batch b
            pending = self.distribute(batches[b + 1], src=src, how=how,
                                      async_op=True) \
                if b + 1 < len(batches) else []
            results.append(engine.remap_tensor(
                self.plan, None, x, [0], mode, threshold=threshold,
                flags=flags, out=None if outs is None else outs[b]))
        return results

    def apply(self, field, remap_axes, mode, threshold=0.0, flags=0,
              tune=None, out=None):
        from pyremap_amd import engine
        return engine.remap_tensor(self.plan, None, field, remap_axes, mode,
                                   threshold=threshold, flags=flags,
                                   tune=tune, out=out)

    def gather(self, y_local, row_axis=0):
        return gather_rows(y_local, self.bounds, row_axis=row_axis,
                           group=self.group)
