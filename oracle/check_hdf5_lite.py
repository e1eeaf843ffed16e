#!/opt/conda/bin/python3.9
"""
Development-time cross-check of pyremap_amd/io/hdf5_lite.py against h5py
(TEST INFRASTRUCTURE ONLY; runs under the image's conda interpreter, the only
one with h5py):  every dataset and attribute of every HDF5 file given on the
command line must read identically through both.

    /opt/conda/bin/python3.9 oracle/check_hdf5_lite.py file.nc [...]
"""
import importlib.util
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location(
    'hdf5_lite', os.path.join(HERE, '..', 'pyremap_amd', 'io',
                              'hdf5_lite.py'))
lite = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lite)


def same(a, b, where):
    if isinstance(b, h5py.Reference):
        assert isinstance(a, lite.Reference), where
        return
    if isinstance(b, (bytes, str)):
        if isinstance(a, np.ndarray):
            a = a[()]
        if isinstance(a, bytes) and isinstance(b, str):
            a = a.decode()
        if isinstance(b, bytes) and isinstance(a, str):
            b = b.decode()
        assert a == b, (where, a, b)
        return
    b = np.asarray(b)
    if b.dtype.kind == 'O':
        flat_b = b.reshape(-1)
        flat_a = a if isinstance(a, list) else [a]
        assert len(flat_a) == len(flat_b), where
        for x, y in zip(flat_a, flat_b):
            if isinstance(y, (bytes, str)):
                same(x, y, where)
            elif isinstance(y, np.ndarray) and y.dtype.kind == 'O':
                assert len(x) == len(y), where     # references
            else:
                np.testing.assert_array_equal(np.asarray(x), np.asarray(y),
                                              err_msg=where)
        return
    if b.dtype.kind == 'V':
        # compound / opaque: h5py's in-memory layout differs from the file's
        assert np.asarray(a).size == b.size, where
        return
    a = np.asarray(a)
    assert a.shape == b.shape, (where, a.shape, b.shape)
    if b.dtype.kind in 'fiu':
        assert a.dtype == b.dtype, (where, a.dtype, b.dtype)
    np.testing.assert_array_equal(a, b, err_msg=where)


def check_group(g_lite, g_h5, path, stats):
    assert sorted(g_lite.keys()) == sorted(g_h5.keys()), \
        (path, g_lite.keys(), list(g_h5.keys()))
    assert sorted(g_lite.attrs) == sorted(g_h5.attrs), \
        (path, list(g_lite.attrs), list(g_h5.attrs))
    for k in g_h5.attrs:
        same(g_lite.attrs[k], g_h5.attrs[k], f'{path}@{k}')
        stats['attrs'] += 1
    for name in g_h5.keys():
        a, b = g_lite[name], g_h5[name]
        where = f'{path}{name}'
        if isinstance(b, h5py.Group):
            assert isinstance(a, lite.Group), where
            check_group(a, b, where + '/', stats)
            continue
        assert sorted(a.attrs) == sorted(b.attrs), \
            (where, list(a.attrs), list(b.attrs))
        for k in b.attrs:
            same(a.attrs[k], b.attrs[k], f'{where}@{k}')
            stats['attrs'] += 1
        if b.shape is None:
            continue
        assert tuple(a.shape) == tuple(b.shape), where
        same(a.read(), b[()], where)
        stats['datasets'] += 1


def main():
    for path in sys.argv[1:]:
        stats = dict(attrs=0, datasets=0)
        with h5py.File(path, 'r') as h5:
            f = lite.File(path)
            check_group(f.root, h5, '/', stats)
            f.close()
        print(f'OK {os.path.basename(path)}: {stats["datasets"]} datasets, '
              f'{stats["attrs"]} attributes identical')


if __name__ == '__main__':
    main()
