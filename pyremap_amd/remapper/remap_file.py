"""
File -> file remapping: what ``Remapper.ncremap`` does in the reference by
running the NCO ``ncremap`` executable (``pyremap/remapper/ncremap.py:15-145``)
is done here as read -> remap on the GPU -> write, with the observable
contract the reference's tests pin (``tests/test_interpolate.py:192-240``):

* the output holds the same data variables as ``remap_numpy`` gives for the
  same input, with equal values (NCO's extra grid variables ``lat_bnds,
  lon_bnds, gw, area, nvertices, lat_vertices, lon_vertices`` -- which the
  reference's tests drop before comparing -- are not written);
* ``renormalize`` is ``remap_numpy``'s ``renormalization_threshold``;
* ``overwrite=False`` and an existing output file is a silent no-op
  (``ncremap.py:18-19``); ``variable_list`` selects data variables
  (``ncremap.py:64-65``); a point-collection source raises ``TypeError``
  (``ncremap.py:20-23``).
"""
import os

import numpy as np

from pyremap_amd.descriptor import (
    MpasMeshDescriptor,
    PointCollectionDescriptor,
)
from pyremap_amd.io import _parallel
from pyremap_amd.io.netcdf import file_format, open_dataset, write_netcdf
from pyremap_amd.io.netcdf3 import FORMATS
from pyremap_amd.remapper.remap_numpy import _remap_numpy
from pyremap_amd.xr_lite import LazyValues

#: MPAS writes this where a field is undefined (below the sea floor, ...)
MPAS_FILL = -9.99999979021476795361e+33

#: Variables of at least this many bytes are STREAMED: read, remapped and
#: written one at a time (the next one travelling meanwhile), as NCO does
#: variable by variable (``pyremap/remapper/ncremap.py:117-145``) -- host
#: memory holds two of them at most, whatever the file.  Smaller ones are
#: handled together (their transfers overlap).  Environment:
#: PYREMAP_AMD_STREAM_BYTES.
STREAM_BYTES = int(os.environ.get('PYREMAP_AMD_STREAM_BYTES', 64 << 20))


def _validate_inputs(remapper, out_filename, overwrite):
    if remapper.map_filename is None:
        raise ValueError('No mapping file has been defined')
    if not overwrite and os.path.exists(out_filename):
        return False  # skip processing
    if isinstance(remapper.src_descriptor, PointCollectionDescriptor):
        raise TypeError(
            'Source grid is a point collection, which is not supported.')
    return True


def _remap_file(remapper, in_filename, out_filename, variable_list,
                overwrite, renormalize, logger, replace_mpas_fill):
    if not _validate_inputs(remapper, out_filename, overwrite):
        return
    group = getattr(remapper, '_process_group', None)
    # (with a variable list only those variables -- and the coordinates --
    # are read: the rest of the file is not touched.  Under
    # Remapper.use_process_group the call is a collective: every rank reads
    # the file and takes part in every variable's remap in the same order,
    # rank `src` alone writes -- no streaming there, the ranks must agree on
    # the order of the collectives)
    ds = open_dataset(in_filename, variables=variable_list,
                      lazy_bytes=None if group is not None else STREAM_BYTES)
    if variable_list is not None:
        missing = [v for v in variable_list if v not in ds]
        if missing:
            raise ValueError(f'variables {missing} are not in {in_filename}')
        ds = ds.drop_vars([v for v in ds.data_vars
                           if v not in variable_list])
    if replace_mpas_fill and isinstance(remapper.src_descriptor,
                                        MpasMeshDescriptor):
        # `ncremap -P mpas` without -C: MPAS's missing value gets the role
        # of _FillValue
        for name in list(ds.data_vars):
            var = ds.variables[name]
            if var.dtype.kind != 'f':
                continue
            fill = np.asarray(MPAS_FILL).astype(var.dtype)
            if var.is_lazy:
                # still on disk: the substitution joins the read
                lazy = var._data

                def load(lazy=lazy, fill=fill):
                    data = lazy.load()
                    if not (data.flags['WRITEABLE'] and
                            data.flags['C_CONTIGUOUS']):
                        data = np.array(data, copy=True, order='C')
                    _parallel.replace_value(data, fill, np.nan)
                    return data
                ds[name] = type(var)(
                    LazyValues(lazy.shape, lazy.dtype, load),
                    dims=var.dims, attrs=var.attrs)
                continue
            hit = var.values == fill
            if hit.any():
                data = np.array(var.values, copy=True)
                data[hit] = np.nan
                ds[name] = type(var)(data, dims=var.dims,
                                     attrs=var.attrs)
    encoding = getattr(ds, 'encoding', {})
    ds_out = _remap_numpy(remapper, ds, renormalize)
    fmt = encoding.get('format') or file_format(in_filename)
    if fmt not in FORMATS and fmt != 'NETCDF4':
        fmt = 'NETCDF3_64BIT_DATA'
    if group is not None and remapper._matrix.rank != group[1]:
        return                       # rank `src` writes
    write_netcdf(ds_out, out_filename, format=fmt,
                 unlimited_dims=encoding.get('unlimited_dims', []))
    if logger is not None:
        logger.info(f'remapped {in_filename} -> {out_filename} with '
                    f'{remapper.map_filename} ({len(list(ds_out.data_vars))} '
                    f'variables)')
