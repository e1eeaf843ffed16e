"""
Multi-GPU application of one mapping: destination rows sharded over the
ranks of a ``torch.distributed`` process group (one process per GPU, backend
``nccl`` = RCCL over xGMI on MI355X nodes; ``gloo`` on CPU for tests).

The path partitions naturally -- destination row i needs only CSR row i and
the source rows it references (``remap_numpy.py:264-268``) -- so:

* every rank holds the CSR rows of ONE contiguous, work-balanced range
  (:func:`row_shard_bounds`) and writes only its slab of Y;
* the only exchange is ONE broadcast of the source field per batch
  (:func:`broadcast_field`); there is no reduction collective;
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

    def broadcast(self, field, src=0):
        return broadcast_field(field, src=src, group=self.group)

    def apply(self, field, remap_axes, mode, threshold=0.0, flags=0,
              tune=None, out=None):
        from pyremap_amd import engine
        return engine.remap_tensor(self.plan, None, field, remap_axes, mode,
                                   threshold=threshold, flags=flags,
                                   tune=tune, out=out)

    def gather(self, y_local, row_axis=0):
        return gather_rows(y_local, self.bounds, row_axis=row_axis,
                           group=self.group)
