"""
A small HDF5 writer (numpy only) that produces NetCDF-4 files: the counterpart
of :mod:`pyremap_amd.io.hdf5_lite`, so that a NetCDF-4 input can be answered
with a NetCDF-4 output (what ``ncremap`` does) on images without netCDF4 /
h5py.

Deliberately the oldest, checksum-free flavour of the format (HDF5 File Format
Specification, version-0 superblock): one root group with a symbol table (one
v1 B-tree leaf, one symbol-table node, a local heap), version-1 object
headers, contiguous little-endian datasets, version-1 attribute messages, one
global heap collection for the object references of ``DIMENSION_LIST``.

The NetCDF-4 data model on top (as netCDF-C writes it): every dimension is a
dimension-scale dataset (``CLASS = "DIMENSION_SCALE"``, ``NAME``,
``_Netcdf4Dimid``) -- a coordinate variable if one of that name exists, else
an unallocated float dataset named "This is a netCDF dimension but not a
netCDF variable." -- and every variable lists its dimensions in
``DIMENSION_LIST`` (one variable-length list holding one object reference per
axis).  Files are checked against libhdf5 (h5py, ``h5dump``) in
``oracle/check_hdf5_write.py``.
"""
import struct
from collections import OrderedDict

import numpy as np

from pyremap_amd.io import _parallel

UNDEF = 0xFFFFFFFFFFFFFFFF
_SIG = b'\x89HDF\r\n\x1a\n'
_PURE_DIM = 'This is a netCDF dimension but not a netCDF variable.'


def _pad8(n):
    return (n + 7) & ~7


# ---------------------------------------------------------------------------
# header messages
# ---------------------------------------------------------------------------

def _datatype(dtype):
    """Datatype message body for a numpy dtype (numbers and S<n> strings)."""
    dt = np.dtype(dtype)
    if dt.kind in 'iu':
        bits = 0x08 if dt.kind == 'i' else 0x00
        return struct.pack('<BBBBI', 0x10, bits, 0, 0, dt.itemsize) + \
            struct.pack('<HH', 0, 8 * dt.itemsize)
    if dt.kind == 'f' and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            sign, prec, eloc, esize, msize, bias = 31, 32, 23, 8, 23, 127
        else:
            sign, prec, eloc, esize, msize, bias = 63, 64, 52, 11, 52, 1023
        return struct.pack('<BBBBI', 0x11, 0x20, sign, 0, dt.itemsize) + \
            struct.pack('<HHBBBBI', 0, prec, eloc, esize, 0, msize, bias)
    if dt.kind == 'S':
        # fixed length, null terminated, ASCII
        return struct.pack('<BBBBI', 0x13, 0x00, 0, 0, dt.itemsize)
    raise TypeError(f'cannot store dtype {dt} in a NetCDF-4 file')


_REF_TYPE = struct.pack('<BBBBI', 0x17, 0x00, 0, 0, 8)        # object ref
_VLEN_REF_TYPE = struct.pack('<BBBBI', 0x19, 0x00, 0, 0, 16) + _REF_TYPE


def _dataspace(shape, unlimited=False):
    rank = len(shape)
    out = struct.pack('<BBB5x', 1, rank, 1 if unlimited and rank else 0)
    out += b''.join(struct.pack('<Q', int(n)) for n in shape)
    if unlimited and rank:
        out += struct.pack('<Q', UNDEF) + b''.join(
            struct.pack('<Q', int(n)) for n in shape[1:])
    return out


def _message(mtype, body, flags=0):
    body = body + b'\x00' * (_pad8(len(body)) - len(body))
    if len(body) > 0xFFF8:
        raise ValueError('a header message exceeds 64 KiB (attribute too '
                         'long for this writer)')
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def _attribute(name, dtype_body, space_body, data):
    nm = name.encode('utf-8') + b'\x00'
    body = struct.pack('<BxHHH', 1, len(nm), len(dtype_body),
                       len(space_body))
    for part in (nm, dtype_body, space_body):
        body += part + b'\x00' * (_pad8(len(part)) - len(part))
    return _message(0x0C, body + data)


def _attr_value(name, value):
    """An attribute message for a python / numpy value."""
    if isinstance(value, bytes):
        value = value.decode('utf-8', 'replace')
    if isinstance(value, str):
        raw = value.encode('utf-8') + b'\x00'
        return _attribute(name, _datatype(f'S{len(raw)}'), _dataspace(()),
                          raw)
    arr = np.asarray(value)
    if arr.dtype.kind in 'US':
        return _attr_value(name, ' '.join(str(v) for v in arr.reshape(-1)))
    if arr.dtype.kind == 'b':
        arr = arr.astype(np.int8)
    if arr.dtype.kind not in 'iuf':
        return _attr_value(name, str(value))
    arr = np.ascontiguousarray(arr.astype(arr.dtype.newbyteorder('<')))
    # numbers are 1-D arrays in netCDF (even single ones)
    return _attribute(name, _datatype(arr.dtype),
                      _dataspace((arr.size,)), arr.tobytes())


def _object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BxHII4x', 1, len(messages), 1, len(body)) + body


# ---------------------------------------------------------------------------
# the file
# ---------------------------------------------------------------------------

class _Dataset:
    def __init__(self, name, data, dims, attrs, allocate=True):
        self.name = name
        self.data = data
        self.dims = tuple(dims)
        self.attrs = attrs
        self.allocate = allocate
        self.header_addr = None
        self.data_addr = None
        self.ref_slots = []       # global-heap object indices, one per axis


def _is_deferred(data):
    return hasattr(data, 'load') and not isinstance(data, np.ndarray)


def write_netcdf4(filename, dimensions, variables, attrs=None,
                  unlimited=(), nan_fill=None, auto_fill=None):
    """
    ``auto_fill``: variable name -> fill value for variables whose data is
    PRODUCED ON DEMAND (an object with ``shape``, ``dtype``, ``load()`` and
    optionally ``prefetch()``): each is loaded when the writer reaches it,
    written and dropped (the next one started meanwhile), and gets that
    value as ``_FillValue`` -- and in place of its NaNs -- iff its values
    hold NaNs.  Their object headers are laid out with the attribute and
    written last; where it turns out not to be needed a NIL message of the
    same size takes its place.


    ``dimensions``: ordered name -> length; ``variables``: iterable of
    ``(name, dims, ndarray, attrs)``; ``attrs``: global attributes;
    ``nan_fill``: variable name -> value stored in place of its NaNs
    (substituted chunk by chunk while writing);
    ``unlimited`` is accepted and ignored: record dimensions are written at
    their current length (HDF5 allows unlimited maxima only with chunked
    storage; this writer stores contiguously).
    """
    attrs = OrderedDict(attrs or {})
    auto_fill = dict(auto_fill or {})
    nan_fill = dict(nan_fill or {})
    dimensions = OrderedDict((k, int(v)) for k, v in dimensions.items())
    datasets = OrderedDict()
    for name, dims, data, vattrs in variables:
        if _is_deferred(data):
            if tuple(data.shape) != tuple(dimensions[d] for d in dims):
                raise ValueError(f'{name}: shape {data.shape} does not '
                                 f'match dimensions {dims}')
            datasets[name] = _Dataset(name, data, dims,
                                      OrderedDict(vattrs or {}))
            continue
        arr = np.asarray(data)
        if arr.dtype.kind == 'U':
            arr = np.char.encode(arr, 'utf-8')
        if arr.dtype.kind == 'b':
            arr = arr.astype(np.int8)
        if arr.dtype.kind in 'iuf':
            # (no copy when the data is little-endian already: a remapped
            # field is as large as memory allows)
            arr = arr.astype(arr.dtype.newbyteorder('<'), copy=False)
        if tuple(arr.shape) != tuple(dimensions[d] for d in dims):
            raise ValueError(f'{name}: shape {arr.shape} does not match '
                             f'dimensions {dims}')
        if arr.ndim:                 # (ascontiguousarray makes 0-d 1-d)
            arr = np.ascontiguousarray(arr)
        datasets[name] = _Dataset(name, arr, dims, OrderedDict(vattrs or {}))
    # dimension scales: coordinate variables, or placeholders
    dim_ids = {d: i for i, d in enumerate(dimensions)}
    for d, n in dimensions.items():
        ds = datasets.get(d)
        if ds is None or ds.dims != (d,):
            if ds is not None:
                raise ValueError(
                    f'variable {d} shares its name with a dimension but is '
                    f'not 1-D along it: NetCDF-4 files written here cannot '
                    f'hold that')
            datasets[d] = _Dataset(d, np.zeros(n, dtype='<f4'), (d,),
                                   OrderedDict(), allocate=False)
    names = sorted(datasets, key=lambda s: s.encode('utf-8'))

    # -- global heap: one 8-byte object per (variable, axis) reference -------
    heap_objects = []          # dimension name of each object, index = i + 1
    for name in names:
        ds = datasets[name]
        if name in dimensions and ds.dims == (name,):
            continue                      # a scale does not list itself
        for d in ds.dims:
            heap_objects.append(d)
            ds.ref_slots.append(len(heap_objects))

    def dataset_messages(ds, gheap_addr, reserve=False):
        # fixed extents: HDF5 allows unlimited maxima only with chunked
        # storage, so record dimensions are written at their current length
        msgs = [_message(0x01, _dataspace(ds.data.shape)),
                _message(0x03, _datatype(np.dtype(ds.data.dtype)
                                         .newbyteorder('<')), flags=0x01)]
        nbytes = ds.data.nbytes
        addr = (ds.data_addr or 0) if (ds.allocate and nbytes) else UNDEF
        msgs.append(_message(0x08, struct.pack('<BBQQ', 3, 1, addr, nbytes)))
        if ds.name in dimensions and ds.dims == (ds.name,):
            label = ds.name if ds.allocate else \
                f'{_PURE_DIM}{dimensions[ds.name]:10d}'
            msgs.append(_attr_value('CLASS', 'DIMENSION_SCALE'))
            msgs.append(_attr_value('NAME', label))
            msgs.append(_attribute('_Netcdf4Dimid', _datatype('<i4'),
                                   _dataspace(()),
                                   struct.pack('<i', dim_ids[ds.name])))
        elif ds.ref_slots:
            data = b''.join(struct.pack('<IQI', 1, gheap_addr, slot)
                            for slot in ds.ref_slots)
            msgs.append(_attribute('DIMENSION_LIST', _VLEN_REF_TYPE,
                                   _dataspace((len(ds.ref_slots),)), data))
        for key, value in ds.attrs.items():
            msgs.append(_attr_value(key, value))
        if _is_deferred(ds.data) and ds.name in auto_fill:
            # the _FillValue of a variable whose values are not known yet:
            # reserved in the layout, written iff NaNs turned up, a NIL
            # message of the same size otherwise
            fill = _attr_value('_FillValue', auto_fill[ds.name])
            if reserve or ds.name in nan_fill:
                msgs.append(fill)
            else:
                msgs.append(_message(0x00, b'\x00' * (len(fill) - 8)))
        return msgs

    # -- pass 1: sizes (addresses do not change any size) --------------------
    leaf_k = max(4, (len(names) + 1) // 2)
    internal_k = 16
    root_msgs = [_message(0x11, struct.pack('<QQ', 0, 0))] + \
        [_attr_value(k, v) for k, v in attrs.items()]
    root_header = _object_header(root_msgs)
    btree_size = 24 + 2 * internal_k * 8 + (2 * internal_k + 1) * 8
    heap_names = [b'']
    heap_names += [n.encode('utf-8') for n in names]
    name_offsets, off = [], 0
    for nm in heap_names:
        name_offsets.append(off)
        off += _pad8(len(nm) + 1)
    heap_data_size = off + 16                    # + one free block
    snod_size = 8 + 2 * leaf_k * 40
    gheap_used = 16 + len(heap_objects) * (16 + 8)
    gheap_size = max(4096, _pad8(gheap_used + 16))

    pos = 96                                     # superblock
    root_addr = pos
    pos += _pad8(len(root_header))
    btree_addr = pos
    pos += btree_size
    heap_addr = pos
    pos += 32
    heap_data_addr = pos
    pos += heap_data_size
    snod_addr = pos
    pos += snod_size
    gheap_addr = pos
    pos += gheap_size
    for name in names:
        ds = datasets[name]
        ds.header_addr = pos
        pos += _pad8(len(_object_header(dataset_messages(ds, gheap_addr,
                                                         reserve=True))))
    for name in names:
        ds = datasets[name]
        if ds.allocate and ds.data.nbytes:
            ds.data_addr = pos
            pos += _pad8(ds.data.nbytes)
    eof = pos

    # -- pass 2: bytes -------------------------------------------------------
    with open(filename, 'wb') as f:
        def put(addr, blob, fill=None):
            f.seek(addr)
            if isinstance(blob, np.ndarray):
                # large arrays: several cores
                _parallel.write_at(f, blob, nan_fill=fill)
            else:
                f.write(blob)

        sb = _SIG + struct.pack('<BBBBBBBB', 0, 0, 0, 0, 0, 8, 8, 0)
        sb += struct.pack('<HHI', leaf_k, internal_k, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
        sb += struct.pack('<QQII', 0, root_addr, 1, 0)
        sb += struct.pack('<QQ', btree_addr, heap_addr)
        put(0, sb)
        root_msgs[0] = _message(0x11, struct.pack('<QQ', btree_addr,
                                                  heap_addr))
        put(root_addr, _object_header(root_msgs))
        # B-tree: one leaf entry -> the symbol-table node
        bt = b'TREE' + struct.pack('<BBHQQ', 0, 0, 1, UNDEF, UNDEF)
        bt += struct.pack('<QQQ', 0, snod_addr, name_offsets[-1])
        put(btree_addr, bt + b'\x00' * (btree_size - len(bt)))
        # local heap: names, then one free block to the end
        put(heap_addr, b'HEAP' + struct.pack('<B3xQQQ', 0, heap_data_size,
                                             heap_data_size - 16,
                                             heap_data_addr))
        seg = bytearray(heap_data_size)
        for nm, o in zip(heap_names, name_offsets):
            seg[o:o + len(nm)] = nm
        seg[heap_data_size - 16:] = struct.pack('<QQ', 1, 16)
        put(heap_data_addr, bytes(seg))
        # symbol-table node, entries sorted by name
        sn = b'SNOD' + struct.pack('<BxH', 1, len(names))
        for name, o in zip(names, name_offsets[1:]):
            sn += struct.pack('<QQII16x', o, datasets[name].header_addr, 0,
                              0)
        put(snod_addr, sn + b'\x00' * (snod_size - len(sn)))
        # global heap collection: the object references
        gh = b'GCOL' + struct.pack('<B3xQ', 1, gheap_size)
        for i, d in enumerate(heap_objects):
            gh += struct.pack('<HH4xQQ', i + 1, 1, 8,
                              datasets[d].header_addr)
        free = gheap_size - len(gh)
        gh += struct.pack('<HH4xQ', 0, 0, free)
        put(gheap_addr, gh + b'\x00' * (gheap_size - len(gh)))
        deferred = [n for n in names if _is_deferred(datasets[n].data)]
        for name in names:
            ds = datasets[name]
            data = ds.data
            if _is_deferred(data):
                later = deferred[deferred.index(name) + 1:]
                if later and hasattr(datasets[later[0]].data, 'prefetch'):
                    datasets[later[0]].data.prefetch()
                arr = np.asarray(data.load())
                if tuple(arr.shape) != tuple(data.shape) or \
                        arr.dtype.itemsize != np.dtype(data.dtype).itemsize:
                    raise ValueError(
                        f'{name}: produced {arr.dtype} {arr.shape}, '
                        f'announced {data.dtype} {tuple(data.shape)}')
                arr = np.ascontiguousarray(
                    arr.astype(arr.dtype.newbyteorder('<'), copy=False))
                if name in auto_fill and _parallel.any_nan(arr):
                    nan_fill[name] = auto_fill[name]
                data = arr
            if ds.data_addr is not None:
                # the array's own memory goes to the file (no tobytes copy)
                put(ds.data_addr, data
                    if data.dtype.kind in 'iuf' and data.size
                    else data.tobytes(), nan_fill.get(name))
            del data
            # (after the data: a deferred variable's _FillValue is known now)
            put(ds.header_addr,
                _object_header(dataset_messages(ds, gheap_addr)))
        f.truncate(eof)
