"""
The package's own HDF5 / NetCDF-4 reader (map and
field files are NetCDF-4 in practice, and the image has neither netCDF4 nor
h5py) against what h5py read from the same files
(tests/golden/make_hdf5_fixtures.py -> tests/golden/hdf5/expected.npz).
CPU only.
"""
import os

import numpy as np
import pytest

from pyremap_amd.io import hdf5_lite, mapfile, netcdf
from pyremap_amd.io.netcdf4_lite import NetCDF4File

HERE = os.path.join(os.path.dirname(__file__), 'golden', 'hdf5')
FILES = {'classic': 'classic.h5', 'latest': 'latest.h5',
         'scales': 'scales.h5', 'map_nc4': 'map_nc4.nc',
         'nc4_mpasAreaVertex': 'nc4_mpasAreaVertex.nc',
         'nc4_ref_latlon_to_mpas_cell': 'nc4_ref_latlon_to_mpas_cell.nc'}


@pytest.fixture(scope='module')
def expected():
    return np.load(os.path.join(HERE, 'expected.npz'))


def _as_expected(value):
    """The normalisation make_hdf5_fixtures.py applied to h5py's values."""
    if isinstance(value, (bytes, str)):
        return np.array(value.decode() if isinstance(value, bytes)
                        else value)
    if isinstance(value, list):
        return np.array([v.decode() if isinstance(v, bytes) else v
                         for v in value])
    return np.asarray(value)


@pytest.mark.parametrize('tag', sorted(FILES))
def test_every_dataset_and_attribute_matches_h5py(expected, tag):
    keys = [k for k in expected.files if k.startswith(tag + ':')]
    assert keys
    checked = 0
    with hdf5_lite.File(os.path.join(HERE, FILES[tag])) as f:
        for key in keys:
            path = key.split(':', 1)[1]
            attr = None
            if '@' in path:
                path, attr = path.split('@')
            obj = f.root
            for part in [p for p in path.split('/') if p]:
                obj = obj[part]
            want = expected[key]
            got = obj.attrs[attr] if attr is not None else obj.read()
            got = _as_expected(got)
            if want.dtype.kind in 'US':
                assert got.astype(str).tolist() == \
                    want.astype(str).tolist(), key
            else:
                assert got.dtype == want.dtype, (key, got.dtype, want.dtype)
                assert got.shape == want.shape, key
                np.testing.assert_array_equal(got, want, err_msg=key)
            checked += 1
    assert checked == len(keys)


def test_format_features_are_the_ones_claimed():
    """The fixtures really exercise both generations of the format."""
    with hdf5_lite.File(os.path.join(HERE, 'classic.h5')) as f:
        assert f.superblock_version == 0
        assert len(f.root['many'].keys()) == 300      # multi-level B-tree
    with hdf5_lite.File(os.path.join(HERE, 'latest.h5')) as f:
        assert f.superblock_version == 3
        assert len(f.root['wide'].keys()) == 400      # dense links
        assert len(f.root['many_attrs'].attrs) == 31  # dense attributes
    with hdf5_lite.File(os.path.join(HERE, 'scales.h5')) as f:
        assert f.base == 512                          # user block


def test_unsupported_input_is_reported(tmp_path):
    bad = tmp_path / 'not.h5'
    bad.write_bytes(b'CDF\x01' + b'\x00' * 100)
    with pytest.raises(ValueError, match='not an HDF5 file'):
        hdf5_lite.File(str(bad))


def test_netcdf4_dimensions_from_scales():
    with NetCDF4File(os.path.join(HERE, 'scales.h5')) as nc:
        assert dict(nc.dimensions) == {'n_s': 6, 'lat': 5, 'lon': 6,
                                       'phony_dim_0': 6}
        assert 'n_s' not in nc.variables          # a dimension, not a var
        assert nc.variables['temp'].dims == ('lat', 'lon')
        assert nc.variables['S'].dims == ('n_s',)
        assert nc.variables['lat'].dims == ('lat',)
        assert nc.variables['loose'].dims == ('phony_dim_0',)
        assert 'DIMENSION_LIST' not in nc.variables['temp'].attrs
        assert 'CLASS' not in nc.variables['lat'].attrs


def test_open_dataset_netcdf4(expected):
    ds = netcdf.open_dataset(
        os.path.join(HERE, 'nc4_ref_latlon_to_mpas_cell.nc'))
    assert ds.encoding['format'] == 'NETCDF4'
    assert ds['SST'].dims == ('time', 'nCells')
    assert ds['SST'].shape == (1, 7153)
    assert ds['SST'].dtype == np.float64
    assert ds['SST'].attrs['units'] == 'degC'
    assert '_FillValue' not in ds['SST'].attrs      # decoded away
    raw = expected['nc4_ref_latlon_to_mpas_cell:/SST']
    np.testing.assert_array_equal(ds['SST'].values, raw)
    assert ds['lat_cell'].attrs['units'] == 'radians'
    assert ds['date'].dtype == np.int32
    assert ds.attrs['title'].strip() != ''
    # -9999 is the declared fill value of areaVertex: none present, so the
    # decoded field equals the stored one
    dv = netcdf.open_dataset(os.path.join(HERE, 'nc4_mpasAreaVertex.nc'))
    np.testing.assert_array_equal(
        dv['areaVertex'].values, expected['nc4_mpasAreaVertex:/areaVertex'])


def test_read_mapping_netcdf4(expected):
    m = mapfile.read_mapping(os.path.join(HERE, 'map_nc4.nc'))
    assert (m.n_a, m.n_b, m.n_s) == (12, 8, 20)
    for name in ('row', 'col', 'S', 'frac_b', 'src_grid_dims',
                 'dst_grid_dims'):
        got = getattr(m, name)
        np.testing.assert_array_equal(got, expected[f'map_nc4:/{name}'])
        assert got.dtype.isnative
    assert m.src_grid_rank == 2 and m.dst_grid_rank == 2
