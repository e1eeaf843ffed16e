#!/opt/conda/bin/python3.9
"""
Development-time check of pyremap_amd/io/hdf5_write.py against libhdf5 (TEST
INFRASTRUCTURE ONLY; runs under the image's conda interpreter, the only one
with h5py): a file written by the package's writer must read back through
h5py value for value, its dimension scales must be scales for the HDF5
dimension-scale API (``H5DSis_scale``) and every variable's axes must
resolve, through ``DIMENSION_LIST``, to the scales the NetCDF-4 model names.

    python -c "...write_netcdf4('/tmp/w.nc', ...); np.savez('/tmp/w.npz', ...)"
    /opt/conda/bin/python3.9 oracle/check_hdf5_write.py /tmp/w.nc /tmp/w.npz
"""
import json
import sys

import h5py
import numpy as np


def main():
    path, expected = sys.argv[1], np.load(sys.argv[2], allow_pickle=False)
    meta = json.loads(str(expected['__meta__']))
    with h5py.File(path, 'r') as f:
        assert sorted(f.keys()) == sorted(meta['datasets']), \
            (sorted(f.keys()), sorted(meta['datasets']))
        for name, dims in meta['variables'].items():
            d = f[name]
            want = expected[f'var/{name}']
            got = d[()]
            assert got.dtype == want.dtype, (name, got.dtype, want.dtype)
            np.testing.assert_array_equal(got, want, err_msg=name)
            if name in meta['dimensions'] and dims == [name]:
                assert h5py.h5ds.is_scale(d.id), name
            else:
                assert len(d.dims) == len(dims)
                for axis, dim in enumerate(dims):
                    scales = [s.name.lstrip('/') for s in d.dims[axis].values()]
                    assert scales == [dim], (name, axis, scales, dim)
            for k, v in meta['attrs'].get(name, {}).items():
                a = d.attrs[k]
                if isinstance(v, str):
                    a = a.decode() if isinstance(a, bytes) else a
                    assert a == v, (name, k, a, v)
                else:
                    np.testing.assert_array_equal(
                        np.asarray(a).reshape(-1), np.asarray(v).reshape(-1))
        for dim, size in meta['dimensions'].items():
            assert h5py.h5ds.is_scale(f[dim].id), dim
            assert f[dim].shape == (size,), dim
            assert int(f[dim].attrs['_Netcdf4Dimid']) == \
                list(meta['dimensions']).index(dim)
        for k, v in meta['global_attrs'].items():
            a = f.attrs[k]
            a = a.decode() if isinstance(a, bytes) else a
            if isinstance(v, str):
                assert a == v, (k, a, v)
            else:
                np.testing.assert_array_equal(np.asarray(a).reshape(-1),
                                              np.asarray(v).reshape(-1))
    print(f'OK {path}: {len(meta["variables"])} variables, '
          f'{len(meta["dimensions"])} dimension scales resolve through '
          f'libhdf5')


if __name__ == '__main__':
    main()
