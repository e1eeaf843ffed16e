"""
In-memory weight application -- the host-side mirror of the reference's
``pyremap/remapper/remap_numpy.py``, with the same function names, argument
meaning and error behaviour, and with the arithmetic (scipy CSR build, the
``matrix.dot`` calls and every numpy pass around them) replaced by the
MI355X engine (:mod:`pyremap_amd.engine` -> ``libremap_hip.so``).

==============================  ============================================
reference (remap_numpy.py)      here
==============================  ============================================
``_remap_numpy``      :19-69    same flow, accepts xarray or xr_lite objects
``_load_mapping``     :72-139   same validation; ``csr_matrix(...)`` becomes
                                ``RemapPlan.from_triplets`` (device CSR)
``_check_drop``       :142-147  identical rule
``_remap_data_array`` :150-220  same dims/coords bookkeeping; ``isnan`` test
                                and mask live on the device
``_remap_numpy_array`` :223-297 ``engine.remap_tensor``: strides instead of
                                transpose copies, one fused HIP launch
==============================  ============================================
"""
import os
import sys

import numpy as np

from pyremap_amd import engine, host_path, xr_lite
from pyremap_amd.io.mapfile import read_mapping

try:  # real xarray is honoured when present; it is optional
    import xarray as _xarray
    if not hasattr(_xarray, '__version__'):
        _xarray = None
except ImportError:  # pragma: no cover - depends on the environment
    _xarray = None


def _is_data_array(obj):
    if isinstance(obj, xr_lite.DataArray):
        return True
    return _xarray is not None and isinstance(obj, _xarray.DataArray)


def _is_dataset(obj):
    if isinstance(obj, xr_lite.Dataset):
        return True
    return _xarray is not None and isinstance(obj, _xarray.Dataset)


def _array_class(like):
    if _xarray is not None and isinstance(like, (_xarray.DataArray,
                                                 _xarray.Dataset)):
        return _xarray.DataArray
    return xr_lite.DataArray


class _MapInfo:
    """What the reference keeps in ``remapper._ds_map`` and reads later."""

    def __init__(self, mapping):
        self.n_a = mapping.n_a
        self.n_b = mapping.n_b
        self.src_grid_rank = mapping.src_grid_rank
        self.dst_grid_rank = mapping.dst_grid_rank
        # grid dimensions are stored in Fortran order (remap_numpy.py:108)
        self.src_grid_dims = [int(d) for d in mapping.src_grid_dims[::-1]]
        self.dst_grid_dims = [int(d) for d in mapping.dst_grid_dims[::-1]]
        self.frac_b = mapping.frac_b


def _remap_numpy(remapper, ds, renormalization_threshold):
    """
    Remap a Dataset or DataArray, possibly masked and renormalized
    (reference ``_remap_numpy`` :19-69: same checks, same errors, same
    ``history`` / ``mesh_name`` attributes).
    """
    if remapper.map_filename is None:
        raise ValueError('No mapping file has been defined')
    _load_mapping(remapper)

    expected = remapper._ds_map.src_grid_dims
    for dim, size in zip(remapper.src_descriptor.dims, expected):
        if size != ds.sizes[dim]:
            raise ValueError(
                f"data set and remapping source dimension {dim} don't "
                f'have the same size: {size} != {ds.sizes[dim]}')

    args = (remapper, renormalization_threshold)
    if _is_data_array(ds):
        result = _remap_data_array(ds, *args)
    elif _is_dataset(ds):
        # variables holding only part of the source dims cannot be remapped
        partial = [name for name in ds.data_vars
                   if _check_drop(remapper, ds[name])]
        kept = ds.drop_vars(partial)
        if any(getattr(kept.variables[name], 'is_lazy', False)
               for name in kept.data_vars):
            # variables still on disk (ncremap's streaming path): the result
            # is lazy too -- each variable is read, remapped and handed to the
            # writer in turn, never all of them at once
            result = _lazy_dataset(remapper, kept, renormalization_threshold)
        else:
            result = kept.map(
                _LookAhead(remapper, kept, list(kept.data_vars),
                           renormalization_threshold), keep_attrs=True)
    else:
        raise TypeError('ds not an xarray Dataset or DataArray.')

    command = ' '.join(sys.argv[:])
    previous = result.attrs.get('history')
    result.attrs['history'] = command if previous is None else \
        '\n'.join([previous, command])
    result.attrs['mesh_name'] = remapper.dst_descriptor.mesh_name
    return result


def _validate_mapping(info, src_descriptor, dst_descriptor):
    """The rank and size checks of reference ``_load_mapping`` :92-132."""
    n_src, n_dst = len(src_descriptor.dims), len(dst_descriptor.dims)
    if n_src != info.src_grid_rank or n_dst != info.dst_grid_rank:
        raise ValueError(
            f'The number of source and/or destination dimensions does not '
            f'match the expected \n'
            f'number of source and destination dimensions in the mapping '
            f'file. \n'
            f'{n_src} != {info.src_grid_rank} and/or {n_dst} != '
            f'{info.dst_grid_rank}')
    sides = (
        ('source mesh descriptor and remapping source dimension',
         src_descriptor, info.src_grid_dims),
        ('dest. mesh descriptor and remapping dest. dimension',
         dst_descriptor, info.dst_grid_dims),
    )
    for label, descriptor, file_dims in sides:
        for dim, have, want in zip(descriptor.dims, descriptor.dim_sizes,
                                   file_dims):
            if have != want:
                raise ValueError(
                    f"{label} {dim} don't have the same size: \n"
                    f'{have} != {want}')


#: device plans of mapping FILES already loaded by this process, most
#: recent last: (path, mtime, size, device, devices) -> (info, plan,
#: schedule).  MPAS-Analysis builds one short-lived Remapper per variable
#: group over the same few mapping files; the second one finds its weights
#: on the device (the reference re-reads and re-sorts per Remapper,
#: remap_numpy.py:87-137).  Environment: PYREMAP_AMD_PLAN_CACHE = number of
#: plans kept (default 4; 0 disables).  Mappings of more than
#: ``_PLAN_CACHE_MAX_NNZ`` entries (their plans hold GBs of HBM) are not kept
#: alive behind the Remapper's back.
_PLAN_CACHE = {}
_PLAN_CACHE_SIZE = int(os.environ.get('PYREMAP_AMD_PLAN_CACHE', 4))
_PLAN_CACHE_MAX_NNZ = 20_000_000


def _plan_cache_key(remapper):
    if remapper._mapping_override is not None or _PLAN_CACHE_SIZE <= 0 or \
            getattr(remapper, '_process_group', None) is not None:
        return None
    try:
        st = os.stat(remapper.map_filename)
    except OSError:
        return None
    devices = getattr(remapper, 'devices', None)
    return (os.path.abspath(remapper.map_filename), st.st_mtime_ns,
            st.st_size, str(remapper.device),
            tuple(str(d) for d in devices) if devices else None)


def _load_mapping(remapper):
    """
    Read the mapping file once, validate it against the descriptors and sort
    its triplets into a device-resident CSR (reference ``_load_mapping``
    :72-139; the plan is cached where the reference caches ``_matrix``, and
    -- per process -- by mapping file, see ``_PLAN_CACHE``).
    """
    if remapper._ds_map is not None:
        return
    key = _plan_cache_key(remapper)
    if key is not None and key in _PLAN_CACHE:
        info, plan, schedule = _PLAN_CACHE.pop(key)
        _PLAN_CACHE[key] = (info, plan, schedule)       # most recent last
        _validate_mapping(info, remapper.src_descriptor,
                          remapper.dst_descriptor)
        remapper._matrix, remapper.schedule = plan, schedule
        remapper._ds_map = info
        return
    mapping = remapper._mapping_override
    if mapping is None:
        mapping = read_mapping(remapper.map_filename)
    info = _MapInfo(mapping)
    _validate_mapping(info, remapper.src_descriptor, remapper.dst_descriptor)
    # csr_matrix((S, (row - 1, col - 1)), shape=(n_b, n_a)), on the device
    plan = engine.RemapPlan.from_triplets(
        mapping.row, mapping.col, mapping.S, mapping.frac_b, info.n_a,
        info.n_b, index_base=1, device=remapper.device)
    devices = getattr(remapper, 'devices', None)
    group = getattr(remapper, '_process_group', None)
    if group is not None:
        # one process per GPU: this rank keeps its rows (packed columns)
        from pyremap_amd.parallel import ShardedRemap
        plan = ShardedRemap(plan, group=group[0],
                            grid_dims=info.dst_grid_dims)
        remapper.schedule = plan.schedule
    elif devices and len(devices) > 1:
        # one process, several GPUs: rows sharded over them
        from pyremap_amd.parallel import MultiDeviceRemap
        plan = MultiDeviceRemap(plan, devices, grid_dims=info.dst_grid_dims)
        remapper.schedule = plan.schedule
    else:
        # pick the kernel schedule for this mapping (LDS-staged patches when
        # neighbouring destination rows share their source rows)
        remapper.schedule = plan.auto_schedule(info.dst_grid_dims)
    remapper._matrix = plan
    remapper._ds_map = info
    if key is not None and getattr(plan, 'nnz', 0) <= _PLAN_CACHE_MAX_NNZ:
        _PLAN_CACHE[key] = (info, plan, remapper.schedule)
        while len(_PLAN_CACHE) > _PLAN_CACHE_SIZE:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))


def _check_drop(remapper, da):
    """True for a variable with some, but not all, source dims (:142-147)."""
    hits = [dim in da.dims for dim in remapper.src_descriptor.dims]
    return any(hits) and not all(hits)


def _remap_data_array(da, remapper, renormalization_threshold):
    """
    Remap one variable (reference ``_remap_data_array`` :150-220): the
    destination dims replace the first source dim, the other source dims
    vanish, coordinates without a source dim are kept and the destination
    descriptor's coordinates are added.
    """
    return _start_data_array(da, remapper, renormalization_threshold)()


def _plan_data_array(da, remapper):
    """
    The bookkeeping of reference ``_remap_data_array`` :150-199 without
    touching the values: ``(remap_axes, dims, coords)`` of the result, or
    ``None`` for a variable without source dims.
    """
    src_dims = remapper.src_descriptor.dims
    remap_axes = [axis for axis, dim in enumerate(da.dims)
                  if dim in src_dims]
    if not remap_axes:
        return None  # nothing to remap
    if len(remap_axes) != len(src_dims):
        raise ValueError(
            'Data array with some (but not all) required source dims cannot '
            'be remapped and should have been dropped.')

    first = remap_axes[0]
    dims = list(da.dims[:first]) + list(remapper.dst_descriptor.dims) + \
        [dim for dim in da.dims[first:] if dim not in src_dims]

    coords = {}
    for name in da.coords:
        coord = da.coords[name]
        if not any(dim in coord.dims for dim in src_dims):
            coords[name] = {'dims': coord.dims, 'data': coord.values}
    coords.update(remapper.dst_descriptor.coords)
    return remap_axes, dims, coords


def _lazy_dataset(remapper, ds, renormalization_threshold):
    """
    ``ds.map(_remap_data_array, keep_attrs=True)`` for a Dataset whose
    variables are (partly) still on disk: same variables, dims, coordinates
    and attributes, but the values of every remapped variable are a
    :class:`~pyremap_amd.xr_lite.LazyValues` -- read, remapped and
    downloaded when the writer asks for them, and started one variable ahead
    by its ``prefetch`` (read, upload and launch: variable i + 1 is computed
    while variable i is being written; its result waits on the DEVICE and
    comes down when asked for, so the host holds one result at a time).
    """
    results = {}
    for name in ds.data_vars:
        da = ds[name]
        plan = _plan_data_array(da, remapper)
        if plan is None:
            out = da
        else:
            remap_axes, dims, coords = plan
            first = remap_axes[0]
            shape = list(da.shape[:first]) + \
                [int(s) for s in remapper._ds_map.dst_grid_dims] + \
                [size for axis, size in enumerate(da.shape)
                 if axis > first and axis not in remap_axes]

            def start(da=da):
                finish = _start_data_array(da, remapper,
                                           renormalization_threshold,
                                           keep_on_device=True)
                return lambda: finish().values
            out = xr_lite.DataArray(
                xr_lite.LazyValues(shape, np.float64,
                                   load=lambda start=start: start()(),
                                   prefetch=start),
                coords={k: xr_lite.DataArray(v['data'], dims=v['dims'],
                                             name=k, attrs=v.get('attrs'))
                        for k, v in coords.items()},
                dims=dims, name=name)
        out.attrs = type(out.attrs)(ds.variables[name].attrs)  # keep_attrs
        results[name] = out
    return xr_lite.Dataset(results, attrs=ds.attrs)


def _start_data_array(da, remapper, renormalization_threshold,
                      keep_on_device=False):
    """
    Enqueue the remap of one variable -- upload, NaN scan, launch, download,
    none of which waits for the host -- and return the function that waits
    for the data and assembles the result.  ``keep_on_device``: the download
    waits too, until the result is asked for (the streaming file path: the
    host then holds one result at a time).
    """
    plan = _plan_data_array(da, remapper)
    if plan is None:
        return lambda: da  # nothing to remap
    remap_axes, dims, coords = plan

    if getattr(remapper, '_process_group', None) is not None:
        data = _collective_array(remapper, da.values, remap_axes,
                                 renormalization_threshold)
        make = _array_class(da).from_dict
        attrs, name = da.attrs, da.name
        return lambda: make({'coords': coords, 'attrs': attrs, 'dims': dims,
                             'data': data, 'name': name})

    # the NaN test of :201-204 and the product both run on the device; masked
    # entries come back as NaN, which is what xarray makes of the reference's
    # masked array
    pending = host_path.remap_host_array(
        remapper._matrix, remapper._ds_map.dst_grid_dims, da.values,
        remap_axes,
        mode='fracb' if renormalization_threshold is None else 'auto',
        threshold=renormalization_threshold, flags=remapper.engine_flags,
        keep_on_device=keep_on_device)
    make = _array_class(da).from_dict
    attrs, name = da.attrs, da.name

    def finish():
        return make({'coords': coords, 'attrs': attrs, 'dims': dims,
                     'data': pending.result(), 'name': name})
    return finish


def _in_memory(da):
    """The numpy array behind ``da`` if its values are already in memory,
    else None (xr_lite's ``LazyValues``; with real xarray a dask array or a
    lazily indexed backend array) -- without touching ``.values``."""
    var = getattr(da, 'variable', da)
    if getattr(var, 'is_lazy', False) or getattr(da, 'is_lazy', False):
        return None
    data = getattr(var, '_data', None)
    return data if isinstance(data, np.ndarray) else None


def _batches(remapper, ds, names):
    """
    ``{name: [names of its batch]}`` for the variables that are remapped
    TOGETHER (``host_path.remap_host_batch``): at least two variables with
    the same dims, shape and dtype, each at most ``BATCH_VAR_BYTES`` --
    a climatology file's dozens of ``(Time, nCells)`` fields.  One device,
    host arrays; everything else keeps the per-variable pipeline.
    """
    plan = getattr(remapper, '_matrix', None)
    if getattr(remapper, '_process_group', None) is not None or \
            plan is None or hasattr(plan, 'shards'):
        return {}
    groups = {}
    src_dims = remapper.src_descriptor.dims
    for name in names:
        da = ds[name]
        # metadata only: `.values` of a dask-backed or lazily indexed
        # variable computes / reads it -- every variable of the Dataset up
        # front, the multi-GB ones that are rejected by size included, and
        # once more when its turn comes.  Only arrays already in memory are
        # batched; the others keep the per-variable pipeline, `depth` at a
        # time.
        data = _in_memory(da)
        if data is None or data.dtype.kind not in 'fiub' or data.size == 0:
            continue
        # what travels: float32 as it is, everything else as float64
        dtype = data.dtype if data.dtype in (np.float32, np.float64) \
            else np.dtype(np.float64)
        nbytes = data.size * dtype.itemsize
        if nbytes > host_path.BATCH_VAR_BYTES:
            continue
        hit = [dim in src_dims for dim in da.dims]
        if sum(hit) != len(src_dims):
            continue
        groups.setdefault((tuple(da.dims), tuple(data.shape), dtype.str),
                          []).append((name, nbytes))
    out = {}
    for sized in groups.values():
        members = [name for name, _ in sized]
        per = max(1, sized[0][1])
        cap = max(2, host_path.BATCH_TOTAL_BYTES // per)
        for i in range(0, len(members), cap):
            part = members[i:i + cap]
            if len(part) >= 2:
                for name in part:
                    out[name] = part
    return out


def _start_batch(ds, names, remapper, renormalization_threshold):
    """``{name: finish}`` as :func:`_start_data_array` gives them, for the
    variables of one batch: one stacked upload / launch / download."""
    first = ds[names[0]]
    remap_axes, _, _ = _plan_data_array(first, remapper)
    pending = host_path.remap_host_batch(
        remapper._matrix, remapper._ds_map.dst_grid_dims,
        [ds[name].values for name in names], remap_axes,
        mode='fracb' if renormalization_threshold is None else 'auto',
        threshold=renormalization_threshold, flags=remapper.engine_flags)
    out = {}
    for index, name in enumerate(names):
        da = ds[name]
        _, dims, coords = _plan_data_array(da, remapper)
        make = _array_class(da).from_dict

        def finish(index=index, dims=dims, coords=coords, make=make,
                   attrs=da.attrs, name=name):
            return make({'coords': coords, 'attrs': attrs, 'dims': dims,
                         'data': pending.result()[index], 'name': name})
        out[name] = finish
    return out


class _LookAhead:
    """
    ``Dataset.map`` calls its function one variable at a time; this keeps the
    next ``depth`` variables' transfers and launches enqueued while one is
    awaited, so a variable's download overlaps its successors' uploads
    (reference: the per-variable loop of ``remap_numpy.py:42-55``).
    """

    def __init__(self, remapper, ds, names, threshold, depth=2):
        self.remapper, self.ds, self.names = remapper, ds, list(names)
        self.threshold, self.depth = threshold, depth
        self.started = {}
        self.position = 0
        self.batches = _batches(remapper, ds, self.names)

    def _start_up_to(self, last):
        while self.position <= min(last, len(self.names) - 1):
            name = self.names[self.position]
            if name not in self.started:
                batch = self.batches.get(name)
                if batch is not None:
                    # small variables of one shape: together (the whole
                    # batch starts with its first member)
                    self.started.update(_start_batch(
                        self.ds, batch, self.remapper, self.threshold))
                else:
                    self.started[name] = _start_data_array(
                        self.ds[name], self.remapper, self.threshold)
            self.position += 1

    def __call__(self, da, *unused):
        name = da.name
        if name not in self.names:
            return _remap_data_array(da, self.remapper, self.threshold)
        self._start_up_to(self.names.index(name) + self.depth)
        return self.started.pop(name)()


def _collective_array(remapper, values, remap_axes, threshold, mode='auto',
                      host_mask=None):
    """
    One array through the process group (``Remapper.use_process_group``):
    rank ``src`` uploads its array, every rank computes its rows from the
    packed source rows it receives, every rank returns the full float64
    result.

    ``mode='auto'`` (what ``_remap_data_array`` wants: NaN where the
    reference masks, masked iff a threshold and a NaN) returns the array;
    ``'masked'`` / ``'fracb'`` (``_remap_numpy_array``) return ``(values,
    mask)`` with the reference's mask of ``remap_numpy.py:278``.
    ``host_mask``: the MaskedArray's mask in masked mode -- burnt in as NaN
    on ``src``, except that an UNMASKED NaN goes through as the reference
    lets it (``:263``: NaN * 1 poisons every cell it touches, which stays
    unmasked), exactly as ``host_path._enqueue`` does on one device.
    """
    torch = engine.require_gpu()
    import torch.distributed as dist
    sharded = remapper._matrix
    src = remapper._process_group[1]
    values = np.asarray(values)
    if values.dtype not in (np.float64, np.float32):
        values = values.astype(np.float64)
    tdtype = torch.float32 if values.dtype == np.float32 else torch.float64
    device = sharded.plan.device
    field = poisoned = None
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    if sharded.rank == src:
        field = torch.from_numpy(np.ascontiguousarray(values)).to(device)
        if host_mask is not None:
            m_d = torch.from_numpy(np.ascontiguousarray(
                host_mask, dtype=np.bool_)).to(device)
            poisoned = torch.isnan(field) & ~m_d
            field.masked_fill_(m_d, float('nan'))
            if bool(poisoned.any()):
                field.masked_fill_(poisoned, 0.0)
                flag[0] = 1
            else:
                poisoned = None
    kw = dict(src=src, flags=remapper.engine_flags,
              shape=tuple(values.shape))
    dims = remapper._ds_map.dst_grid_dims
    if mode == 'auto':
        y = sharded.remap_tensor(dims, field, remap_axes,
                                 threshold=threshold, dtype=tdtype, **kw)
        return y.cpu().numpy()
    y, mask = sharded.remap_tensor(
        dims, field, remap_axes, threshold=threshold, dtype=tdtype,
        mode=mode, want_mask=True, **kw)
    if host_mask is not None:
        if sharded.world_size > 1:
            dist.broadcast(flag, src=src, group=sharded.group)
        if int(flag[0]):
            # 0 * NaN = NaN through a RAW product marks every destination
            # cell a poisoned entry touches
            p_field = None
            if sharded.rank == src:
                p_field = torch.where(poisoned, float('nan'), 0.0).to(
                    torch.float64)
            hit = sharded.remap_tensor(dims, p_field, remap_axes,
                                       dtype=torch.float64, mode='raw', **kw)
            y = torch.where(torch.isnan(hit), float('nan'), y)
    return y.cpu().numpy(), mask.cpu().numpy().astype(bool)


def _remap_numpy_array(remapper, in_field, remap_axes,
                       renormalization_threshold):
    """
    Remap a single array (reference ``_remap_numpy_array`` :223-297).

    ``in_field`` may be

    * a ``numpy.ndarray`` or ``numpy.ma.MaskedArray`` -- the result is a
      ``numpy.ma.MaskedArray`` with the reference's shape, values and mask
      (masked mode iff the input is a MaskedArray and a threshold is given,
      exactly as at :258-261; the data under the mask is NaN here, the
      undivided sum in the reference -- not observable through the API);
    * a ``torch.Tensor`` on the plan's device -- the result is a float64
      device tensor holding NaN where the reference masks; masked mode is
      chosen as ``_remap_data_array`` would (a threshold and at least one
      NaN), nothing leaves the device.
    """
    torch = engine.require_gpu()
    plan = remapper._matrix
    dst_grid_dims = remapper._ds_map.dst_grid_dims
    remap_axes = [int(a) for a in remap_axes]

    if getattr(remapper, '_process_group', None) is not None:
        # collective: rank src's data, every rank gets the full result
        is_ma = isinstance(in_field, np.ma.MaskedArray)
        if isinstance(in_field, torch.Tensor):
            src = remapper._process_group[1]
            return plan.remap_tensor(
                dst_grid_dims,
                in_field.to(plan.plan.device) if plan.rank == src else None,
                remap_axes, threshold=renormalization_threshold, src=src,
                flags=remapper.engine_flags, shape=tuple(in_field.shape),
                dtype=in_field.dtype if in_field.dtype in (
                    torch.float32, torch.float64) else torch.float64)
        # the same branch, stand-ins and output mask as on one device (below)
        masked = is_ma and renormalization_threshold is not None
        data = np.ma.getdata(in_field) if is_ma else np.asarray(in_field)
        host_mask = np.ma.getmaskarray(in_field) if masked else None
        out, mask = _collective_array(
            remapper, data, remap_axes,
            renormalization_threshold if masked else None,
            mode='masked' if masked else 'fracb', host_mask=host_mask)
        return np.ma.masked_array(out, mask=mask)

    if isinstance(in_field, torch.Tensor):
        field = in_field.to(plan.device)
        if field.dtype not in (torch.float64, torch.float32):
            field = field.to(torch.float64)
        if renormalization_threshold is None:
            return engine.remap_tensor(plan, dst_grid_dims, field,
                                       remap_axes, engine.MODE_FRACB,
                                       flags=remapper.engine_flags)
        # NaN scan and both candidate launches stay on the device
        return engine.remap_tensor_auto_mode(
            plan, dst_grid_dims, field, remap_axes,
            renormalization_threshold, flags=remapper.engine_flags)

    is_ma = isinstance(in_field, np.ma.MaskedArray)
    masked = is_ma and renormalization_threshold is not None
    data = np.ma.getdata(in_field) if is_ma else np.asarray(in_field)
    host_mask = None
    if masked:
        # the kernel reads the mask off NaNs (what in_mask * in_field amounts
        # to when the mask is isnan(field), :201-204 and :263): the mask goes
        # up beside the data and is burnt in on the device
        host_mask = np.ma.getmaskarray(in_field)
        if not host_mask.any() and not host_path._any_nan(
                np.ascontiguousarray(data)):
            # nothing masked and no NaN an empty mask would have to protect
            # (an UNMASKED NaN goes through in the reference, :263)
            host_mask = None
    out, mask = host_path.remap_host_array(
        plan, dst_grid_dims, data, remap_axes,
        mode='masked' if masked else 'fracb',
        threshold=renormalization_threshold if masked else None,
        want_mask=True, flags=remapper.engine_flags,
        host_mask=host_mask).result()
    return np.ma.masked_array(out, mask=mask)
