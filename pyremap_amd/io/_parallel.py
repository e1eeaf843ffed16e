"""
Large array <-> file traffic on several cores.

Byte-order conversion and page-cache copies run at one core's memory speed
(~3 GB/s); a remapped field is gigabytes, and the file -> file path
(``Remapper.ncremap``) spends nine tenths of its time in exactly these two
loops.  numpy releases the GIL inside its copy / cast loops and ``os.pwrite``
/ ``os.preadv`` inside the system call, so plain threads scale.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

#: arrays below this many bytes are not worth a thread pool
MIN_BYTES = 32 << 20
#: bytes handled per task
CHUNK_BYTES = 16 << 20


def _workers():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(8, n))


def _spans(n_items, itemsize):
    step = max(1, CHUNK_BYTES // max(itemsize, 1))
    return [(i, min(i + step, n_items)) for i in range(0, n_items, step)]


def convert(src, dtype):
    """
    ``src.astype(dtype)`` (C-contiguous result) with the element loop spread
    over threads; ``src`` may be a memory map in file byte order.
    """
    src = np.asarray(src)
    dtype = np.dtype(dtype)
    if src.nbytes < MIN_BYTES or _workers() == 1 or \
            not src.flags['C_CONTIGUOUS']:
        return np.ascontiguousarray(src.astype(dtype))
    out = np.empty(src.shape, dtype=dtype)
    s, o = src.reshape(-1), out.reshape(-1)

    def task(span):
        i, j = span
        o[i:j] = s[i:j]

    with ThreadPoolExecutor(_workers()) as pool:
        list(pool.map(task, _spans(s.size, max(src.dtype.itemsize,
                                                dtype.itemsize))))
    return out


def any_nan(data):
    """``np.isnan(data).any()`` chunk by chunk with an early exit (no
    full-size boolean temporary)."""
    data = np.asarray(data)
    if data.dtype.kind != 'f' or data.size == 0:
        return False
    flat = data.reshape(-1) if data.flags['C_CONTIGUOUS'] else data.ravel()
    step = max(1, CHUNK_BYTES // data.dtype.itemsize)
    for i in range(0, flat.size, step):
        if np.isnan(flat[i:i + step]).any():
            return True
    return False


def replace_value(data, old, new):
    """``data[data == old] = new`` in place (C-contiguous ``data``), chunk
    by chunk and on several cores for large arrays."""
    flat = data.reshape(-1)

    def task(span):
        i, j = span
        part = flat[i:j]
        part[part == old] = new

    spans = _spans(flat.size, data.dtype.itemsize)
    if data.nbytes < MIN_BYTES or _workers() == 1:
        for span in spans:
            task(span)
    else:
        with ThreadPoolExecutor(_workers()) as pool:
            list(pool.map(task, spans))


def pread_convert(fd, runs, file_dtype, count):
    """
    ``count`` elements of ``file_dtype`` (any byte order) from a file into a
    fresh NATIVE-order array: ``runs`` = ``(file offset, first element,
    elements)`` triples -- one for a contiguous variable, one per record for
    a record variable.  Every run is cut into pieces of a few MB; each task
    ``preadv``s its piece into a small buffer of its own and casts it into
    place (both release the GIL: several cores).  Unlike a memory map of the
    file -- whose touched pages stay resident beside the converted copy --
    nothing but the result remains, which is what a streaming remap of a
    large file needs.
    """
    file_dtype = np.dtype(file_dtype)
    native = file_dtype.newbyteorder('=')
    out = np.empty(count, dtype=native)
    item = file_dtype.itemsize
    step = max(1, (8 << 20) // item)
    pieces = []
    for off, first, n in runs:
        for i in range(0, n, step):
            pieces.append((off + i * item, first + i, min(step, n - i)))
    same = file_dtype.byteorder in '=|' or file_dtype == native or \
        item == 1

    def task(piece):
        off, first, n = piece
        if same:
            mv = memoryview(out[first:first + n].view(np.uint8))
        else:
            buf = np.empty(n, dtype=file_dtype)
            mv = memoryview(buf.view(np.uint8))
        done, want = 0, n * item
        while done < want:
            got = os.preadv(fd, [mv[done:]], off + done)
            if got <= 0:
                raise OSError('short read')
            done += got
        if not same:
            out[first:first + n] = buf

    if len(pieces) <= 1 or _workers() == 1:
        for piece in pieces:
            task(piece)
    else:
        with ThreadPoolExecutor(_workers()) as pool:
            list(pool.map(task, pieces))
    return out


def write_at(f, data, dtype=None, nan_fill=None):
    """
    Write the C-order bytes of ``data`` (converted to ``dtype`` first if
    given, e.g. big-endian; NaNs stored as ``nan_fill`` if given) at the
    current position of the binary file ``f`` and advance it; large arrays go
    out as parallel ``pwrite`` calls, each task converting its own chunk into
    its own small buffer -- no full-size temporary at any point.  Returns
    the number of bytes written.
    """
    data = np.asarray(data)
    dtype = data.dtype if dtype is None else np.dtype(dtype)
    flat = data.reshape(-1) if data.flags['C_CONTIGUOUS'] else \
        np.ascontiguousarray(data).reshape(-1)
    nbytes = flat.size * dtype.itemsize
    if nbytes == 0:
        return 0
    same = nan_fill is None and dtype == flat.dtype and \
        dtype.byteorder in ('=', '|', flat.dtype.byteorder)

    def converted(i, j):
        buf = np.empty(j - i, dtype=dtype)
        buf[:] = flat[i:j]
        if nan_fill is not None:
            buf[np.isnan(flat[i:j])] = nan_fill
        return buf

    if nbytes < MIN_BYTES or _workers() == 1 or not hasattr(os, 'pwrite'):
        step = max(1, CHUNK_BYTES // dtype.itemsize)
        for i in range(0, flat.size, step):
            j = min(i + step, flat.size)
            f.write((flat[i:j] if same else converted(i, j)).view(np.uint8))
        return nbytes
    f.flush()
    fd = f.fileno()
    start = f.tell()

    def task(span):
        i, j = span
        raw = (flat[i:j] if same else converted(i, j)).view(np.uint8)
        mv = memoryview(raw)
        off = start + i * dtype.itemsize
        done = 0
        while done < len(mv):
            done += os.pwrite(fd, mv[done:], off + done)

    with ThreadPoolExecutor(_workers()) as pool:
        list(pool.map(task, _spans(flat.size, dtype.itemsize)))
    f.seek(start + nbytes)
    return nbytes
