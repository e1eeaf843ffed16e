"""
Self-contained reader / writer for the NetCDF classic formats:
CDF-1 (``NETCDF3_CLASSIC``), CDF-2 (``NETCDF3_64BIT`` offsets) and CDF-5
(``NETCDF3_64BIT_DATA``, the reference's default for SCRIP files,
``pyremap/remapper/remapper.py:129``).

The reference does all file IO through xarray + netCDF4
(``remap_numpy.py:88``, ``utility.py:53-67``); neither exists on the MI355X
boxes, and the NCO tools behind ``ncremap`` do not either, so the file ->
file path of this package (``remapper/remap_file.py``) reads and writes
classic-format files itself.  Only numpy is needed.

Format reference: the NetCDF classic format specification (header =
magic, numrecs, dim_list, gatt_list, var_list; big-endian; values padded to
4 bytes; record variables interleaved per record).
"""
import os
import struct
from collections import OrderedDict

import numpy as np

from pyremap_amd.io import _parallel

NC_DIMENSION = 10
NC_VARIABLE = 11
NC_ATTRIBUTE = 12

# nc_type -> big-endian numpy dtype
_TYPES = {
    1: np.dtype('>i1'), 2: np.dtype('S1'), 3: np.dtype('>i2'),
    4: np.dtype('>i4'), 5: np.dtype('>f4'), 6: np.dtype('>f8'),
    # CDF-5 only
    7: np.dtype('>u1'), 8: np.dtype('>u2'), 9: np.dtype('>u4'),
    10: np.dtype('>i8'), 11: np.dtype('>u8'),
}
_CODES = {
    ('i', 1): 1, ('S', 1): 2, ('i', 2): 3, ('i', 4): 4, ('f', 4): 5,
    ('f', 8): 6, ('u', 1): 7, ('u', 2): 8, ('u', 4): 9, ('i', 8): 10,
    ('u', 8): 11,
}

FORMATS = {'NETCDF3_CLASSIC': 1, 'NETCDF3_64BIT': 2,
           'NETCDF3_64BIT_OFFSET': 2, 'NETCDF3_64BIT_DATA': 5}


def _pad4(n):
    return (n + 3) // 4 * 4


class Variable:
    """A variable of a classic-format file."""

    def __init__(self, name, dims, data, attrs=None, is_record=False,
                 nan_fill=None, auto_fill=None):
        self.name = name
        self.dims = tuple(dims)
        self.data = data
        self.attrs = OrderedDict(attrs) if attrs else OrderedDict()
        self.is_record = is_record
        #: when writing: the value stored in place of NaNs (the writer
        #: substitutes it chunk by chunk; ``data`` keeps its NaNs)
        self.nan_fill = nan_fill
        #: when writing a variable whose ``data`` is produced on demand (an
        #: object with ``shape``, ``dtype``, ``load()`` and optionally
        #: ``prefetch()``): the fill value it gets -- as ``_FillValue`` and
        #: in place of its NaNs -- IF its values turn out to hold NaNs
        self.auto_fill = auto_fill

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def shape(self):
        return self.data.shape


class NetCDF3File:
    """Dimensions, global attributes and variables of one file."""

    def __init__(self):
        self.version = 2
        self.dimensions = OrderedDict()   # name -> length (None = record)
        self.attrs = OrderedDict()
        self.variables = OrderedDict()
        self.numrecs = 0


# ---------------------------------------------------------------------------
# reading
# ---------------------------------------------------------------------------

class _Reader:
    def __init__(self, buf, version):
        self.buf = buf
        self.pos = 4
        self.wide = version == 5

    def u32(self):
        v = struct.unpack_from('>I', self.buf, self.pos)[0]
        self.pos += 4
        return v

    def nonneg(self):
        if self.wide:
            v = struct.unpack_from('>Q', self.buf, self.pos)[0]
            self.pos += 8
            return v
        return self.u32()

    def name(self):
        n = self.nonneg()
        s = bytes(self.buf[self.pos:self.pos + n]).decode('utf-8')
        self.pos += _pad4(n)
        return s

    def values(self, nc_type, n):
        dt = _TYPES[nc_type]
        nbytes = n * dt.itemsize
        raw = np.frombuffer(self.buf, dtype=dt, count=n, offset=self.pos)
        self.pos += _pad4(nbytes)
        return raw

    def attrs(self):
        tag = self.u32()
        n = self.nonneg()
        out = OrderedDict()
        if tag == 0:
            return out
        if tag != NC_ATTRIBUTE:
            raise ValueError('corrupt NetCDF header (attribute list)')
        for _ in range(n):
            name = self.name()
            nc_type = self.u32()
            nelems = self.nonneg()
            vals = self.values(nc_type, nelems)
            if nc_type == 2:
                # NUL terminators written by C programs are not part of the
                # text (netCDF4-python drops them as well)
                out[name] = vals.tobytes().decode(
                    'utf-8', 'replace').replace('\x00', '')
            else:
                vals = vals.astype(vals.dtype.newbyteorder('='))
                out[name] = vals[0] if nelems == 1 else vals
        return out


class Deferred:
    """A variable's data not read yet: shape and dtype (native byte order)
    are known from the header, :meth:`read` reads that variable alone."""

    def __init__(self, filename, name, shape, dtype):
        self.filename, self.name = filename, name
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * \
            self.dtype.itemsize

    def read(self):
        if self.layout is not None:
            return _read_direct(self.filename, self.shape, self.dtype,
                                *self.layout)
        return read(self.filename, variables={self.name}) \
            .variables[self.name].data

    #: (begin, record stride or 0): where the bytes lie, for the direct read
    layout = None


def _read_direct(filename, shape, dtype, begin, recsize):
    """One variable's bytes straight into its final, native-order array
    (``_parallel.pread_convert``), not through a memory map of the file."""
    count = int(np.prod(shape, dtype=np.int64))
    if recsize:
        per = count // shape[0] if shape[0] else 0
        runs = [(begin + r * recsize, r * per, per) for r in range(shape[0])]
    else:
        runs = [(begin, 0, count)]
    fd = os.open(filename, os.O_RDONLY)
    try:
        out = _parallel.pread_convert(fd, runs, dtype.newbyteorder('>'),
                                      count)
    finally:
        os.close(fd)
    return out.reshape(shape)


def read(filename, variables=None, defer_bytes=None):
    """
    Read a classic-format file into a :class:`NetCDF3File`: every variable,
    or -- ``variables`` given -- only those named plus the coordinate
    variables (a variable that is 1-D along the dimension of its own name);
    the others are not touched on disk.  With ``defer_bytes`` the data of
    non-coordinate variables of at least that many bytes is left on disk:
    their ``data`` is a :class:`Deferred`.
    """
    buf = np.memmap(filename, dtype=np.uint8, mode='r')
    if bytes(buf[:3]) != b'CDF' or buf[3] not in (1, 2, 5):
        raise ValueError(f'{filename}: not a NetCDF classic (CDF-1/2/5) file')
    nc = NetCDF3File()
    nc.version = int(buf[3])
    rd = _Reader(buf, nc.version)
    numrecs = rd.nonneg()
    streaming = numrecs == (0xFFFFFFFFFFFFFFFF if rd.wide else 0xFFFFFFFF)

    tag = rd.u32()
    n = rd.nonneg()
    dim_names = []
    if tag == NC_DIMENSION:
        for _ in range(n):
            name = rd.name()
            length = rd.nonneg()
            nc.dimensions[name] = None if length == 0 else int(length)
            dim_names.append(name)
    elif tag != 0:
        raise ValueError('corrupt NetCDF header (dimension list)')
    nc.attrs = rd.attrs()

    tag = rd.u32()
    n = rd.nonneg()
    headers = []
    if tag == NC_VARIABLE:
        for _ in range(n):
            name = rd.name()
            ndims = rd.nonneg()
            dimids = [rd.nonneg() for _ in range(ndims)]
            attrs = rd.attrs()
            nc_type = rd.u32()
            vsize = rd.nonneg()
            if nc.version == 1:
                begin = rd.u32()
            else:
                begin = struct.unpack_from('>Q', buf, rd.pos)[0]
                rd.pos += 8
            headers.append((name, dimids, attrs, nc_type, vsize, begin))
    elif tag != 0:
        raise ValueError('corrupt NetCDF header (variable list)')

    def is_rec(dimids):
        return len(dimids) > 0 and nc.dimensions[dim_names[dimids[0]]] is None

    rec_vars = [h for h in headers if is_rec(h[1])]
    if len(rec_vars) == 1:
        h = rec_vars[0]
        shape = [nc.dimensions[dim_names[d]] for d in h[1][1:]]
        recsize = int(np.prod(shape, dtype=np.int64)) * _TYPES[h[3]].itemsize
    else:
        recsize = sum(h[4] for h in rec_vars)
    if streaming:
        if rec_vars and recsize:
            first = min(h[5] for h in rec_vars)
            numrecs = (len(buf) - first) // recsize
        else:
            numrecs = 0
    nc.numrecs = int(numrecs)

    for name, dimids, attrs, nc_type, vsize, begin in headers:
        dims = [dim_names[d] for d in dimids]
        if variables is not None and name not in variables and \
                dims != [name]:
            continue
        dt = _TYPES[nc_type]
        if defer_bytes is not None and dims != [name] and dt.kind != 'S':
            shape = [nc.numrecs if nc.dimensions[d] is None
                     else nc.dimensions[d] for d in dims]
            later = Deferred(filename, name, shape, dt.newbyteorder('='))
            later.layout = (int(begin), int(recsize) if is_rec(dimids) else 0)
            if later.nbytes >= defer_bytes:
                nc.variables[name] = Variable(name, dims, later, attrs,
                                              is_rec(dimids))
                continue
        if is_rec(dimids):
            inner = [nc.dimensions[d] for d in dims[1:]]
            count = int(np.prod(inner, dtype=np.int64))
            if nc.numrecs == 0:
                data = np.zeros([0] + inner, dtype=dt)
            else:
                # strided view over the interleaved records
                base = np.frombuffer(buf, dtype=np.uint8, offset=begin)
                rows = np.lib.stride_tricks.as_strided(
                    base, shape=(nc.numrecs, count * dt.itemsize),
                    strides=(recsize, 1), writeable=False)
                data = np.ascontiguousarray(rows).view(dt).reshape(
                    [nc.numrecs] + inner)
            rec = True
        else:
            shape = [nc.dimensions[d] for d in dims]
            count = int(np.prod(shape, dtype=np.int64)) if shape else 1
            data = np.frombuffer(buf, dtype=dt, count=count,
                                 offset=begin).reshape(shape)
            rec = False
        if dt.kind != 'S':
            data = _parallel.convert(data, dt.newbyteorder('='))
        else:
            data = np.array(data)
        nc.variables[name] = Variable(name, dims, data, attrs, rec)
    return nc


# ---------------------------------------------------------------------------
# writing
# ---------------------------------------------------------------------------

class _Writer:
    def __init__(self, version):
        self.parts = []
        self.wide = version == 5
        self.version = version

    def u32(self, v):
        self.parts.append(struct.pack('>I', v))

    def nonneg(self, v):
        self.parts.append(struct.pack('>Q' if self.wide else '>I', v))

    def offset(self, v):
        self.parts.append(struct.pack('>I' if self.version == 1 else '>Q', v))

    def name(self, s):
        b = s.encode('utf-8')
        self.nonneg(len(b))
        self.parts.append(b + b'\x00' * (_pad4(len(b)) - len(b)))

    def attrs(self, attrs):
        if not attrs:
            self.u32(0)
            self.nonneg(0)
            return
        self.u32(NC_ATTRIBUTE)
        self.nonneg(len(attrs))
        for key, value in attrs.items():
            self.name(key)
            if isinstance(value, bytes):
                value = value.decode('utf-8', 'replace')
            if isinstance(value, str):
                raw = value.encode('utf-8')
                self.u32(2)
                self.nonneg(len(raw))
            else:
                arr = np.atleast_1d(np.asarray(value))
                if arr.dtype == np.bool_:
                    arr = arr.astype(np.int8)
                arr = _storable(arr, self.version)
                self.u32(_code(arr.dtype, self.version))
                self.nonneg(arr.size)
                raw = arr.astype(arr.dtype.newbyteorder('>')).tobytes()
            self.parts.append(raw + b'\x00' * (_pad4(len(raw)) - len(raw)))

    def size(self):
        return sum(len(p) for p in self.parts)


def _code(dtype, version):
    key = (dtype.kind if dtype.kind != 'U' else 'S', dtype.itemsize)
    if dtype.kind == 'S':
        key = ('S', 1)
    if key not in _CODES:
        raise TypeError(f'dtype {dtype} cannot be stored in a NetCDF-3 file')
    code = _CODES[key]
    if code > 6 and version != 5:
        raise TypeError(f'dtype {dtype} needs NETCDF3_64BIT_DATA (CDF-5)')
    return code


def _storable(arr, version):
    """Map dtypes the target version lacks onto the closest stored type."""
    if arr.dtype.kind == 'b':
        return arr.astype(np.int8)
    if version != 5:
        if arr.dtype.kind == 'i' and arr.dtype.itemsize == 8:
            return arr.astype(np.int32) if np.all(
                np.abs(arr) < 2 ** 31) else arr.astype(np.float64)
        if arr.dtype.kind == 'u':
            return arr.astype({1: np.int16, 2: np.int32}.get(
                arr.dtype.itemsize, np.float64))
    return arr


def _write_big_endian(f, data, nan_fill=None):
    """
    Write ``data`` in C order as big-endian bytes through small reusable
    buffers (a remapped field is as large as memory allows:
    ``astype('>f8').tobytes()`` would hold two more copies of it), large
    arrays on several cores.  Returns the number of bytes written.
    """
    data = np.asarray(data)
    if data.dtype.kind == 'S' or data.dtype.itemsize == 1:
        return _parallel.write_at(f, data)
    return _parallel.write_at(f, data, data.dtype.newbyteorder('>'),
                              nan_fill=nan_fill)


def _is_deferred(data):
    return hasattr(data, 'load') and not isinstance(data, np.ndarray)


def write(filename, dimensions, variables, attrs=None, version=2):
    """
    Write a classic-format file.

    dimensions : dict name -> length (``None`` for the record dimension)
    variables  : iterable of :class:`Variable`; a variable whose first
                 dimension is the record dimension becomes a record variable

    A variable whose ``data`` is produced on demand (``load()``, see
    :attr:`Variable.auto_fill`) is loaded when the writer reaches it, written
    and dropped, the next one being started meanwhile (``prefetch()``): a
    file of many large variables streams through with one or two of them in
    memory.  Whether such a variable needs a ``_FillValue`` is known only
    then, so the header is laid out WITH that attribute for each of them
    (its largest form), the data goes to the offsets that layout gives, and
    the header is written last, in its final form; it may end before the
    first variable begins -- the classic format addresses data by the
    explicit ``begin`` offsets, free space behind the header is legal.
    """
    dimensions = OrderedDict(dimensions)
    dim_ids = {name: i for i, name in enumerate(dimensions)}
    rec_dim = next((n for n, length in dimensions.items() if length is None),
                   None)
    prepared = []
    numrecs = 0
    for var in variables:
        if _is_deferred(var.data):
            data = var.data          # shape / dtype only, for now
        else:
            data = np.asarray(var.data)
            if data.dtype.kind == 'U':
                data = np.char.encode(data, 'utf-8')
            if data.dtype.kind == 'S' and data.dtype.itemsize != 1:
                raise TypeError(
                    f'{var.name}: store strings as S1 char arrays')
            data = _storable(data, version)
        rec = len(var.dims) > 0 and var.dims[0] == rec_dim
        if rec:
            numrecs = max(numrecs, data.shape[0])
        prepared.append((var, data, rec))

    rec_list = [p for p in prepared if p[2]]

    def vsize(data, rec):
        shape = data.shape[1:] if rec else data.shape
        n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        return n * data.dtype.itemsize

    single_rec = len(rec_list) == 1

    def build(begins, reserve):
        w = _Writer(version)
        w.parts.append(b'CDF' + bytes([version]))
        w.nonneg(numrecs)
        if dimensions:
            w.u32(NC_DIMENSION)
            w.nonneg(len(dimensions))
            for name, length in dimensions.items():
                w.name(name)
                w.nonneg(0 if length is None else int(length))
        else:
            w.u32(0)
            w.nonneg(0)
        w.attrs(attrs or {})
        if prepared:
            w.u32(NC_VARIABLE)
            w.nonneg(len(prepared))
            for (var, data, rec), begin in zip(prepared, begins):
                w.name(var.name)
                w.nonneg(len(var.dims))
                for d in var.dims:
                    w.nonneg(dim_ids[d])
                vattrs = var.attrs
                if reserve and _is_deferred(data) and \
                        var.auto_fill is not None:
                    vattrs = OrderedDict(var.attrs)
                    vattrs['_FillValue'] = var.auto_fill
                w.attrs(vattrs)
                w.u32(_code(data.dtype, version))
                w.nonneg(_pad4(vsize(data, rec)))
                w.offset(begin)
        else:
            w.u32(0)
            w.nonneg(0)
        return w

    header = build([0] * len(prepared), True)
    reserved = header.size()
    pos = reserved
    begins = [0] * len(prepared)
    for idx, (var, data, rec) in enumerate(prepared):
        if not rec:
            begins[idx] = pos
            pos += _pad4(vsize(data, rec))
    rec_start = pos
    recsize = 0
    for idx, (var, data, rec) in enumerate(prepared):
        if rec:
            begins[idx] = rec_start + recsize
            recsize += vsize(data, rec) if single_rec else \
                _pad4(vsize(data, rec))
    eof = rec_start + recsize * numrecs if rec_list else pos

    def materialise(idx):
        """Load a deferred variable; decide its _FillValue; start the next."""
        var, data, rec = prepared[idx]
        for later in range(idx + 1, len(prepared)):
            nxt = prepared[later][1]
            if _is_deferred(nxt):
                if hasattr(nxt, 'prefetch'):
                    nxt.prefetch()
                break
        if not _is_deferred(data):
            return data
        arr = np.asarray(data.load())
        if tuple(arr.shape) != tuple(data.shape) or \
                arr.dtype.itemsize != data.dtype.itemsize:
            raise ValueError(
                f'{var.name}: produced {arr.dtype} {arr.shape}, announced '
                f'{data.dtype} {tuple(data.shape)}')
        if var.auto_fill is not None and _parallel.any_nan(arr):
            var.nan_fill = var.auto_fill
            var.attrs['_FillValue'] = var.auto_fill
        return arr

    with open(filename, 'wb') as f:
        f.truncate(eof)
        # non-record variables, in order
        for idx, (var, data, rec) in enumerate(prepared):
            if rec:
                continue
            arr = materialise(idx)
            f.seek(begins[idx])
            n = _write_big_endian(f, arr, var.nan_fill)
            f.write(b'\x00' * (_pad4(n) - n))
            del arr
        # record variables: records interleave the variables, so a
        # variable's records go to their strided places one by one
        for idx, (var, data, rec) in enumerate(prepared):
            if not rec:
                continue
            arr = materialise(idx)
            per = vsize(arr, True)
            if arr.shape[0] and per < (1 << 20):
                # many short records (time stamps, scalars per step): one
                # strided assignment through a mapping of the file instead
                # of a seek and a write per record
                f.flush()
                raw = np.ascontiguousarray(arr)
                if var.nan_fill is not None:
                    raw = np.where(np.isnan(raw), var.nan_fill, raw)
                if raw.dtype.kind != 'S' and raw.dtype.itemsize > 1:
                    raw = raw.astype(raw.dtype.newbyteorder('>'))
                rows = raw.reshape(arr.shape[0], -1).view(np.uint8)
                mm = np.memmap(filename, dtype=np.uint8, mode='r+')
                np.lib.stride_tricks.as_strided(
                    mm[begins[idx]:], shape=rows.shape,
                    strides=(recsize, 1))[:] = rows
                mm.flush()
                del mm
            else:
                for r in range(arr.shape[0]):
                    f.seek(begins[idx] + r * recsize)
                    _write_big_endian(f, arr[r:r + 1], var.nan_fill)
            del arr
        header = build(begins, False)
        if header.size() > reserved:
            raise RuntimeError('NetCDF header outgrew its reserved space')
        f.seek(0)
        f.write(b''.join(header.parts))
