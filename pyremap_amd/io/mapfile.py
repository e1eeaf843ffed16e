"""
Reading and writing SCRIP/ESMF-style mapping ("weights") files.

The reference opens the mapping file with ``xr.open_dataset``
(``pyremap/remapper/remap_numpy.py:88``) and consumes the dims ``n_a, n_b,
src_grid_rank, dst_grid_rank`` and the variables ``src_grid_dims,
dst_grid_dims, col, row, S, frac_b`` (:89-137, :270).
Here the same members are read without xarray:

* NetCDF-3 (CDF-1 / CDF-2 / CDF-5) with :mod:`pyremap_amd.io.netcdf3`;
* NetCDF-4 / HDF5 (what ESMF writes with ``--netcdf4``,
  ``build_map.py:166``) with :mod:`pyremap_amd.io.netcdf4_lite`;
* ``.npz`` with the same variable names (handy for synthetic maps).
"""
import os
from collections import OrderedDict

import numpy as np

REQUIRED = ('src_grid_dims', 'dst_grid_dims', 'col', 'row', 'S', 'frac_b')


def _native(a):
    a = np.asarray(a)
    return np.ascontiguousarray(a, dtype=a.dtype.newbyteorder('='))


class MappingFile:
    """The members of a mapping file that the remapping path consumes."""

    def __init__(self, n_a, n_b, src_grid_dims, dst_grid_dims, row, col, S,
                 frac_b):
        self.n_a = int(n_a)
        self.n_b = int(n_b)
        #: as stored in the file: Fortran order
        self.src_grid_dims = np.asarray(src_grid_dims, dtype=np.int64)
        self.dst_grid_dims = np.asarray(dst_grid_dims, dtype=np.int64)
        #: 1-based, unsorted, duplicates allowed (native byte order)
        self.row = _native(row)
        self.col = _native(col)
        self.S = np.asarray(S, dtype=np.float64)
        self.frac_b = np.asarray(frac_b, dtype=np.float64)

    @property
    def src_grid_rank(self):
        return int(self.src_grid_dims.shape[0])

    @property
    def dst_grid_rank(self):
        return int(self.dst_grid_dims.shape[0])

    @property
    def n_s(self):
        return int(self.S.shape[0])


def _magic(filename):
    with open(filename, 'rb') as f:
        return f.read(8)


def read_mapping(filename):
    """Read the members listed above from ``filename``."""
    if not os.path.exists(filename):
        raise FileNotFoundError(filename)
    magic = _magic(filename)
    if magic[:2] == b'PK' or filename.endswith('.npz'):
        return _read_npz(filename)
    if magic[:3] == b'CDF':
        return _read_netcdf3(filename)
    if magic == b'\x89HDF\r\n\x1a\n':
        return _read_hdf5(filename)
    raise ValueError(f'{filename}: not a NetCDF, HDF5 or npz mapping file')


def _read_npz(filename):
    with np.load(filename) as z:
        missing = [k for k in REQUIRED if k not in z]
        if missing:
            raise ValueError(f'{filename}: missing variables {missing}')
        n_b = int(z['n_b']) if 'n_b' in z else int(z['frac_b'].shape[0])
        if 'n_a' in z:
            n_a = int(z['n_a'])
        else:
            n_a = int(np.prod(z['src_grid_dims']))
        return MappingFile(n_a, n_b, z['src_grid_dims'], z['dst_grid_dims'],
                           z['row'], z['col'], z['S'], z['frac_b'])


def _read_netcdf3(filename):
    from pyremap_amd.io import netcdf3
    nc = netcdf3.read(filename)
    missing = [k for k in REQUIRED if k not in nc.variables]
    if missing:
        raise ValueError(f'{filename}: missing variables {missing}')
    get = {k: nc.variables[k].data for k in REQUIRED}
    for dim in ('n_a', 'n_b'):
        if dim not in nc.dimensions:
            raise ValueError(f'{filename}: missing dimension {dim}')
    return MappingFile(nc.dimensions['n_a'], nc.dimensions['n_b'],
                       get['src_grid_dims'], get['dst_grid_dims'],
                       get['row'], get['col'], get['S'], get['frac_b'])


def _read_hdf5(filename):
    """NetCDF-4 mapping files (ESMF with ``--netcdf4``, ``build_map.py:166``)
    through this package's own HDF5 reader."""
    from pyremap_amd.io.netcdf4_lite import NetCDF4File
    with NetCDF4File(filename) as nc:
        missing = [k for k in REQUIRED if k not in nc.variables]
        if missing:
            raise ValueError(f'{filename}: missing variables {missing}')
        get = {k: _native(nc.variables[k].read()) for k in REQUIRED}
        n_b = nc.dimensions.get('n_b', get['frac_b'].shape[0])
        n_a = nc.dimensions.get('n_a')
        if n_a is None:
            n_a = int(np.prod(get['src_grid_dims']))
    return MappingFile(n_a, n_b, get['src_grid_dims'], get['dst_grid_dims'],
                       get['row'], get['col'], get['S'], get['frac_b'])


def write_mapping(filename, n_a, n_b, src_grid_dims, dst_grid_dims, row, col,
                  S, frac_b, attrs=None, format=None):
    """
    Write a mapping file with the schema of SURVEY.md Appendix A.
    ``*.npz`` -> numpy archive; otherwise ``format`` is ``'NETCDF4'`` (what
    ``ESMF_RegridWeightGen --netcdf4`` writes, ``build_map.py:166``), or a
    classic format; the default is NetCDF-3 64-bit offset, switching to the
    64-bit-data flavour when a variable outgrows 4 GiB.
    ``src_grid_dims`` / ``dst_grid_dims`` are in FILE (Fortran) order and
    ``row`` / ``col`` are 1-based, exactly as ESMF writes them.
    """
    row = np.asarray(row, dtype=np.int32)
    col = np.asarray(col, dtype=np.int32)
    S = np.asarray(S, dtype=np.float64)
    frac_b = np.asarray(frac_b, dtype=np.float64)
    src_grid_dims = np.asarray(src_grid_dims, dtype=np.int32)
    dst_grid_dims = np.asarray(dst_grid_dims, dtype=np.int32)
    if filename.endswith('.npz'):
        np.savez(filename, n_a=np.int64(n_a), n_b=np.int64(n_b),
                 src_grid_dims=src_grid_dims, dst_grid_dims=dst_grid_dims,
                 row=row, col=col, S=S, frac_b=frac_b)
        return
    from pyremap_amd.io import netcdf3
    dims = OrderedDict([
        ('n_a', int(n_a)), ('n_b', int(n_b)), ('n_s', int(S.shape[0])),
        ('src_grid_rank', int(src_grid_dims.shape[0])),
        ('dst_grid_rank', int(dst_grid_dims.shape[0]))])
    variables = [
        ('src_grid_dims', ('src_grid_rank',), src_grid_dims),
        ('dst_grid_dims', ('dst_grid_rank',), dst_grid_dims),
        ('col', ('n_s',), col),
        ('row', ('n_s',), row),
        ('S', ('n_s',), S),
        ('frac_b', ('n_b',), frac_b),
        # keeps n_a a used dimension, as in ESMF files
        ('area_a', ('n_a',), np.zeros(int(n_a))),
    ]
    if format is None:
        big = max(v[2].nbytes for v in variables) >= (1 << 32) - 4
        format = 'NETCDF3_64BIT_DATA' if big else 'NETCDF3_64BIT'
    if format in ('NETCDF4', 'NETCDF4_CLASSIC'):
        from pyremap_amd.io.hdf5_write import write_netcdf4
        write_netcdf4(filename, dims, [v + ({},) for v in variables],
                      attrs=attrs or {})
        return
    if format not in netcdf3.FORMATS:
        raise ValueError(f'unknown mapping-file format {format!r}')
    netcdf3.write(filename, dims,
                  [netcdf3.Variable(*v) for v in variables],
                  attrs=attrs or {}, version=netcdf3.FORMATS[format])
