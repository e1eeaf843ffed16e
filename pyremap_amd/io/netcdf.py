"""
Dataset-level NetCDF IO for the file -> file path, without xarray/netCDF4.

``open_dataset`` plays the role of ``xarray.open_dataset`` (CF mask-and-scale
decoding: ``_FillValue`` / ``missing_value`` become NaN, ``scale_factor`` /
``add_offset`` are applied) and ``write_netcdf`` that of the reference's
``pyremap.utility.write_netcdf`` (``utility.py:8-72``): a ``_FillValue`` is
written only for numeric variables that actually hold NaNs, with netCDF4's
default fill value for the dtype (:38-51).

Classic formats (CDF-1/2/5) are handled by :mod:`pyremap_amd.io.netcdf3`,
NetCDF-4/HDF5 by :mod:`pyremap_amd.io.netcdf4_lite` (read) and
:mod:`pyremap_amd.io.hdf5_write` (write): this package's own HDF5 code, as the
images have neither netCDF4 nor h5py.
"""
import os
from collections import OrderedDict

import numpy as np

from pyremap_amd import xr_lite
from pyremap_amd.io import _parallel, netcdf3

#: netCDF4.default_fillvals (netCDF4 is not installed here)
DEFAULT_FILLVALS = {
    'S1': b'\x00', 'i1': -127, 'u1': 255, 'i2': -32767, 'u2': 65535,
    'i4': -2147483647, 'u4': 4294967295, 'i8': -9223372036854775806,
    'u8': 18446744073709551614, 'f4': 9.969209968386869e+36,
    'f8': 9.969209968386869e+36,
}

_HDF5_MAGIC = b'\x89HDF\r\n\x1a\n'


def file_format(filename):
    with open(filename, 'rb') as f:
        magic = f.read(8)
    if magic[:3] == b'CDF':
        return {1: 'NETCDF3_CLASSIC', 2: 'NETCDF3_64BIT',
                5: 'NETCDF3_64BIT_DATA'}.get(magic[3])
    if magic == _HDF5_MAGIC:
        return 'NETCDF4'
    return None


def _decode(data, attrs):
    """CF mask-and-scale decoding, as xarray applies by default."""
    attrs = OrderedDict(attrs)
    encoding = {}
    if data.dtype.kind not in 'fiu':
        return data, attrs, encoding
    fills = []
    for key in ('_FillValue', 'missing_value'):
        if key in attrs:
            encoding[key] = attrs.pop(key)
            fills.extend(np.atleast_1d(encoding[key]).tolist())
    scale = attrs.pop('scale_factor', None)
    offset = attrs.pop('add_offset', None)
    if fills or scale is not None or offset is not None:
        if data.dtype.kind != 'f':
            data = data.astype(np.float32 if data.dtype.itemsize <= 2
                               else np.float64)
        elif not (data.flags['WRITEABLE'] and data.flags['C_CONTIGUOUS']):
            data = np.array(data, copy=True)
        # (else: the readers hand over fresh arrays nobody else holds)
        for fv in fills:
            fv = np.asarray(fv).astype(data.dtype)
            if not np.isnan(fv):
                _parallel.replace_value(data, fv, np.nan)
        if scale is not None:
            encoding['scale_factor'] = scale
            data = data * np.asarray(scale, dtype=data.dtype)
        if offset is not None:
            encoding['add_offset'] = offset
            data = data + np.asarray(offset, dtype=data.dtype)
    return data, attrs, encoding


def _decoded_dtype(dtype, attrs):
    """The dtype :func:`_decode` gives data of ``dtype`` with ``attrs``."""
    dtype = np.dtype(dtype)
    if dtype.kind not in 'iu':
        return dtype
    if any(k in attrs for k in ('_FillValue', 'missing_value',
                                'scale_factor', 'add_offset')):
        return np.dtype(np.float32 if dtype.itemsize <= 2 else np.float64)
    return dtype


def _lazy_variable(name, dims, shape, dtype, attrs, read, mask_and_scale):
    """A DataArray whose values are read (and decoded) when asked for."""
    if mask_and_scale:
        # attrs / encoding as _decode leaves them, without touching the data
        _, out_attrs, enc = _decode(np.zeros(0, dtype=dtype), attrs)
        out_dtype = _decoded_dtype(dtype, attrs)
    else:
        out_attrs, enc, out_dtype = OrderedDict(attrs), {}, np.dtype(dtype)

    def load():
        data = read()
        if mask_and_scale:
            data = _decode(data, attrs)[0]
        return data
    da = xr_lite.DataArray(xr_lite.LazyValues(shape, out_dtype, load),
                           dims=dims, name=name, attrs=out_attrs)
    da.encoding = enc
    return da, enc


def open_dataset(filename, mask_and_scale=True, variables=None,
                 lazy_bytes=None):
    """
    Read ``filename`` into a :class:`pyremap_amd.Dataset`.  ``variables``:
    read only these data variables (and the coordinate variables); the rest
    of the file is not touched.  ``lazy_bytes``: data variables of at least
    that many bytes stay on disk until their values are asked for
    (:class:`pyremap_amd.xr_lite.LazyValues`: every access reads again).
    """
    if not os.path.exists(filename):
        raise FileNotFoundError(filename)
    fmt = file_format(filename)
    if fmt is None:
        raise ValueError(f'{filename}: not a NetCDF file')
    if variables is not None:
        variables = set(variables)
    if fmt == 'NETCDF4':
        return _open_hdf5(filename, mask_and_scale, variables, lazy_bytes)
    nc = netcdf3.read(filename, variables=variables, defer_bytes=lazy_bytes)
    ds = xr_lite.Dataset(attrs=nc.attrs)
    ds.encoding = {
        'format': fmt,
        'unlimited_dims': [n for n, length in nc.dimensions.items()
                           if length is None],
        'dim_order': list(nc.dimensions),
    }
    for name, var in nc.variables.items():
        if isinstance(var.data, netcdf3.Deferred):
            da, enc = _lazy_variable(name, var.dims, var.data.shape,
                                     var.data.dtype, var.attrs,
                                     var.data.read, mask_and_scale)
            ds[name] = da
            ds.variables[name].encoding = enc
            continue
        data, attrs, enc = (var.data, var.attrs, {})
        if mask_and_scale:
            data, attrs, enc = _decode(data, attrs)
        da = xr_lite.DataArray(data, dims=var.dims, name=name, attrs=attrs)
        da.encoding = enc
        if var.dims == (name,):
            ds._set_coord(name, da)
        else:
            ds[name] = da
        ds.variables[name].encoding = enc
    return ds


def _read_hdf5_variable(filename, name):
    from pyremap_amd.io.netcdf4_lite import NetCDF4File
    with NetCDF4File(filename) as nc:
        return nc.variables[name].read()


def _open_hdf5(filename, mask_and_scale, variables=None, lazy_bytes=None):
    """NetCDF-4 through this package's own HDF5 reader (no h5py/netCDF4)."""
    from pyremap_amd.io.netcdf4_lite import NetCDF4File
    with NetCDF4File(filename) as nc:
        ds = xr_lite.Dataset(attrs=nc.attrs)
        ds.encoding = {'format': 'NETCDF4',
                       'unlimited_dims': list(nc.unlimited),
                       'dim_order': list(nc.dimensions)}
        for name, var in nc.variables.items():
            if variables is not None and name not in variables and \
                    tuple(var.dims) != (name,):
                continue
            try:
                dt = np.dtype(var.dtype)
            except TypeError:          # variable-length strings and the like
                dt = None
            if lazy_bytes is not None and tuple(var.dims) != (name,) and \
                    dt is not None and dt.kind in 'fiu' and \
                    int(np.prod(var.shape, dtype=np.int64)) * dt.itemsize \
                    >= lazy_bytes:
                da, enc = _lazy_variable(
                    name, var.dims, var.shape, dt.newbyteorder('='),
                    OrderedDict(var.attrs),
                    lambda n=name: _read_hdf5_variable(filename, n),
                    mask_and_scale)
                ds[name] = da
                ds.variables[name].encoding = enc
                continue
            data = var.read()
            if isinstance(data, list):
                # variable-length strings: an object array of str
                data = np.array(data, dtype=object).reshape(var.shape)
            attrs, enc = OrderedDict(var.attrs), {}
            if mask_and_scale:
                data, attrs, enc = _decode(data, attrs)
            da = xr_lite.DataArray(data, dims=var.dims, name=name,
                                   attrs=attrs)
            da.encoding = enc
            if var.dims == (name,):
                ds._set_coord(name, da)
            else:
                ds[name] = da
            ds.variables[name].encoding = enc
    return ds


def write_netcdf(ds, filename, format='NETCDF3_64BIT', fillvalues=None,
                 unlimited_dims=None):
    """
    Write a Dataset with netCDF4-style fill values (reference
    ``utility.write_netcdf``): numeric variables holding NaNs get
    ``_FillValue`` = the default fill of their dtype and have their NaNs
    stored as that value; all other variables get no ``_FillValue``.
    """
    nc4 = format in ('NETCDF4', 'NETCDF4_CLASSIC')
    if not nc4 and format not in netcdf3.FORMATS:
        raise NotImplementedError(
            f'format {format!r}: expected NETCDF4, NETCDF4_CLASSIC or one '
            f'of the classic formats {sorted(netcdf3.FORMATS)}')
    version = None if nc4 else netcdf3.FORMATS[format]
    if fillvalues is None:
        fillvalues = DEFAULT_FILLVALS
    if unlimited_dims is None:
        unlimited_dims = getattr(ds, 'encoding', {}).get('unlimited_dims', [])
    names = list(ds.data_vars) + list(ds.coords)
    dimensions = OrderedDict()
    out_vars = []
    for name in names:
        var = ds.variables[name]
        attrs = OrderedDict((k, v) for k, v in var.attrs.items()
                            if k != '_FillValue')
        if getattr(var, 'is_lazy', False) and var._data.dtype.kind == 'f':
            # values produced when the writer reaches the variable: whether
            # they hold NaNs -- hence whether the variable gets a _FillValue
            # -- is found out THEN (the writers reserve the attribute's room
            # in the header and write the header last)
            lazy = var._data
            key = f'{lazy.dtype.kind}{lazy.dtype.itemsize}'
            auto = np.asarray(fillvalues[key]).astype(lazy.dtype) \
                if lazy.dtype.kind == 'f' and key in fillvalues else None
            for dim, size in zip(var.dims, lazy.shape):
                if dim in unlimited_dims:
                    dimensions.setdefault(dim, None)
                else:
                    dimensions.setdefault(dim, int(size))
            out_vars.append(netcdf3.Variable(name, tuple(var.dims), lazy,
                                             attrs, auto_fill=auto))
            continue
        data = np.asarray(var.values)
        nan_fill = None
        if data.dtype.kind == 'f' and _parallel.any_nan(data):
            key = f'{data.dtype.kind}{data.dtype.itemsize}'
            if key in fillvalues:
                # the writers substitute the fill value chunk by chunk: no
                # full-size mask or copy of a field as large as memory allows
                fv = np.asarray(fillvalues[key]).astype(data.dtype)
                nan_fill = fv
                attrs['_FillValue'] = fv
        if data.dtype.kind in 'SU' and data.dtype.itemsize != 1:
            # fixed-width strings -> char array with a string dimension
            raw = np.char.encode(data, 'utf-8') if data.dtype.kind == 'U' \
                else data
            width = raw.dtype.itemsize
            data = raw.reshape(raw.shape + (1,)).view('S1').reshape(
                raw.shape + (width,))
            dims = tuple(var.dims) + (f'string{width}',)
        else:
            dims = tuple(var.dims)
        for dim, size in zip(dims, data.shape):
            if dim in unlimited_dims:
                dimensions.setdefault(dim, None)
            else:
                dimensions.setdefault(dim, int(size))
        out_vars.append(netcdf3.Variable(name, dims, data, attrs,
                                         nan_fill=nan_fill))
    # the record dimension must lead its variables; demote it otherwise
    for dim in [d for d, length in dimensions.items() if length is None]:
        if any(dim in v.dims[1:] for v in out_vars):
            size = next(v.data.shape[v.dims.index(dim)] for v in out_vars
                        if dim in v.dims)
            dimensions[dim] = int(size)
    if sum(1 for length in dimensions.values() if length is None) > 1:
        first = True
        for dim, length in list(dimensions.items()):
            if length is None:
                if not first:
                    size = next(v.data.shape[v.dims.index(dim)]
                                for v in out_vars if dim in v.dims)
                    dimensions[dim] = int(size)
                first = False
    attrs = OrderedDict()
    for key, value in ds.attrs.items():
        attrs[key] = value if isinstance(value, (str, bytes, np.ndarray,
                                                 int, float, np.generic)) \
            else str(value)
    if nc4:
        from pyremap_amd.io.hdf5_write import write_netcdf4
        sizes = OrderedDict()
        for var in out_vars:
            for dim, size in zip(var.dims, var.data.shape):
                sizes.setdefault(dim, int(size))
        write_netcdf4(filename, sizes,
                      [(v.name, v.dims, v.data, v.attrs) for v in out_vars],
                      attrs=attrs,
                      unlimited=[d for d, n in dimensions.items()
                                 if n is None],
                      nan_fill={v.name: v.nan_fill for v in out_vars
                                if v.nan_fill is not None},
                      auto_fill={v.name: v.auto_fill for v in out_vars
                                 if v.auto_fill is not None})
        return
    netcdf3.write(filename, dimensions, out_vars, attrs=attrs,
                  version=version)
