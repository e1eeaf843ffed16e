"""
The NetCDF-4 data model on top of :mod:`pyremap_amd.io.hdf5_lite`: which HDF5
datasets are variables, what their dimensions are called, which attributes are
the user's.

NetCDF-4 stores every dimension as an HDF5 *dimension scale* dataset
(``CLASS = "DIMENSION_SCALE"``); a dimension without a coordinate variable is a
scale whose ``NAME`` starts with "This is a netCDF dimension but not a netCDF
variable".  A variable names its dimensions through ``DIMENSION_LIST``: one
variable-length list of object references per axis, each pointing at the
scale's object header.  Files written by plain h5py without scales get
``phony_dim_N`` names by size, as the netCDF library does.
"""
from collections import OrderedDict

import numpy as np

from pyremap_amd.io import hdf5_lite

_INTERNAL = ('DIMENSION_LIST', 'REFERENCE_LIST', 'CLASS', 'NAME',
             '_Netcdf4Dimid', '_Netcdf4Coordinates', '_nc3_strict',
             '_NCProperties')
_PURE_DIM = 'This is a netCDF dimension but not a netCDF variable'


def _text(value):
    if isinstance(value, bytes):
        return value.decode('utf-8', 'replace')
    if isinstance(value, np.ndarray) and value.dtype.kind == 'S':
        if value.ndim == 0:
            return value[()].decode('utf-8', 'replace')
        if value.size == 1:
            return value.reshape(-1)[0].decode('utf-8', 'replace')
    if isinstance(value, np.bytes_):
        return bytes(value).decode('utf-8', 'replace')
    return value


def _attr(value):
    """An HDF5 attribute value as netCDF4/xarray would hand it over."""
    value = _text(value)
    if isinstance(value, list):
        if len(value) == 1 and isinstance(value[0], str):
            return value[0]
        return value
    if isinstance(value, np.ndarray) and value.dtype.kind in 'fiu' and \
            value.size == 1:
        return value.reshape(-1)[0]
    return value


class Variable:
    def __init__(self, name, dims, dataset, attrs):
        self.name = name
        self.dims = tuple(dims)
        self.attrs = attrs
        self._dataset = dataset
        self.shape = tuple(dataset.shape or ())
        self.dtype = dataset.dtype

    def read(self):
        return self._dataset.read()


class NetCDF4File:
    """``dimensions`` (name -> size), ``unlimited`` (names), ``variables``
    (name -> :class:`Variable`) and global ``attrs`` of the root group."""

    def __init__(self, filename):
        self._h5 = hdf5_lite.File(filename)
        root = self._h5.root
        self.attrs = OrderedDict(
            (k, _attr(v)) for k, v in root.attrs.items()
            if k not in _INTERNAL)
        self.dimensions = OrderedDict()
        self.unlimited = []
        self.variables = OrderedDict()
        datasets = OrderedDict()
        for name in root.keys():
            try:
                obj = root[name]
            except NotImplementedError:
                continue
            if isinstance(obj, hdf5_lite.Dataset):
                datasets[name] = obj
        by_address = {}
        scales = []
        for name, obj in datasets.items():
            cls = _text(obj.attrs.get('CLASS', b''))
            if cls == 'DIMENSION_SCALE':
                by_address[obj.address] = name
                dimid = obj.attrs.get('_Netcdf4Dimid')
                order = int(np.asarray(dimid).reshape(-1)[0]) \
                    if dimid is not None else len(scales)
                scales.append((order, name, obj))
        for _, name, obj in sorted(scales, key=lambda s: s[0]):
            size = obj.shape[0] if obj.shape else 1
            self.dimensions[name] = int(size)
            if obj.maxshape and obj.maxshape[0] == (1 << 64) - 1:
                self.unlimited.append(name)
        phony = {}
        for name, obj in datasets.items():
            attrs = obj.attrs
            label = _text(attrs.get('NAME', b''))
            is_scale = name in self.dimensions
            if is_scale and isinstance(label, str) and \
                    label.startswith(_PURE_DIM):
                continue
            shape = obj.shape or ()
            dims = []
            dimlist = attrs.get('DIMENSION_LIST')
            if dimlist is not None and len(dimlist) == len(shape):
                for axis, refs in enumerate(dimlist):
                    target = refs[0].address if len(refs) else None
                    dims.append(by_address.get(target))
            elif is_scale and len(shape) == 1:
                dims = [name]
            else:
                dims = [None] * len(shape)
            for axis, dim in enumerate(dims):
                if dim is None:
                    size = int(shape[axis])
                    if size not in phony:
                        phony[size] = f'phony_dim_{len(phony)}'
                        self.dimensions[phony[size]] = size
                    dims[axis] = phony[size]
            user = OrderedDict((k, _attr(v)) for k, v in attrs.items()
                               if k not in _INTERNAL)
            self.variables[name] = Variable(name, dims, obj, user)
        # an unlimited dimension is as long as its longest variable (the
        # scale dataset of a coordinate-less record dimension stays empty)
        for var in self.variables.values():
            for dim, size in zip(var.dims, var.shape):
                if dim in self.unlimited and size > self.dimensions[dim]:
                    self.dimensions[dim] = int(size)

    def close(self):
        self._h5.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
