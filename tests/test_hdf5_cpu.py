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


def test_netcdf4_writer_round_trip(tmp_path):
    """``io/hdf5_write.py``: what it writes, the reader reads back -- values,
    dtypes, dimension names (through ``DIMENSION_LIST`` object references in
    the global heap), coordinate variables as dimension scales, placeholder
    scales for dimensions without a variable, attributes of every kind."""
    from collections import OrderedDict

    from pyremap_amd.io.hdf5_write import write_netcdf4
    rng = np.random.default_rng(0)
    dims = OrderedDict([('time', 2), ('lat', 3), ('lon', 4), ('nchar', 5),
                        ('big', 300)])
    label = np.array([list(b'hello'), list(b'world')],
                     dtype='u1').view('S1').reshape(2, 5)
    variables = [
        ('lat', ('lat',), np.linspace(-60, 60, 3), {'units': 'degrees_north'}),
        ('lon', ('lon',), np.linspace(0, 270, 4).astype('>f8'), {}),
        ('temp', ('time', 'lat', 'lon'),
         rng.standard_normal((2, 3, 4)).astype('f4'),
         {'units': 'K', '_FillValue': np.float32(9.96921e36),
          'valid_range': np.array([-5.0, 5.0], 'f4'),
          'flag_values': np.array([1, 2, 3], 'i1')}),
        ('count', ('time',), np.array([3, 4], 'i4'), {}),
        ('wide', ('big', 'lon'), rng.integers(0, 1 << 40, (300, 4)), {}),
        ('u16', ('lat',), np.array([1, 2, 65535], 'u2'), {}),
        ('scalar', (), np.float64(2.5), {'long_name': 'a scalar variable'}),
        ('zeros', ('time', 'lat'), np.zeros((2, 3)), {}),
        ('label', ('time', 'nchar'), label, {}),
    ]
    for i in range(60):           # many links: one wide symbol-table node
        variables.append((f'v{i:02d}', ('lat',), rng.random(3), {}))
    attrs = OrderedDict([('title', 'written here'),
                         ('history', 'line 1\nline 2'),
                         ('version', np.int32(3)),
                         ('scale', 0.5), ('levels', [1, 2, 3])])
    path = str(tmp_path / 'w.nc')
    write_netcdf4(path, dims, variables, attrs=attrs, unlimited=['time'])
    with NetCDF4File(path) as nc:
        assert dict(nc.dimensions) == dict(dims)
        assert list(nc.dimensions) == list(dims)          # _Netcdf4Dimid
        for name in ('time', 'nchar', 'big'):
            assert name not in nc.variables               # pure dimensions
        for name, vdims, data, vattrs in variables:
            var = nc.variables[name]
            assert var.dims == tuple(vdims), name
            got = var.read()
            want = np.asarray(data)
            assert got.dtype == want.dtype.newbyteorder('='), name
            np.testing.assert_array_equal(got, want, err_msg=name)
            for k, v in vattrs.items():
                np.testing.assert_array_equal(np.asarray(var.attrs[k]),
                                              np.asarray(v), err_msg=k)
        assert nc.attrs['title'] == 'written here'
        assert nc.attrs['history'] == 'line 1\nline 2'
        assert nc.attrs['version'] == 3 and nc.attrs['scale'] == 0.5
        assert np.asarray(nc.attrs['levels']).tolist() == [1, 2, 3]
    with hdf5_lite.File(path) as f:
        assert f.superblock_version == 0
        assert len(f.root.keys()) == len(variables) + 3
    # a variable named like a dimension must be that dimension's coordinate
    with pytest.raises(ValueError, match='shares its name'):
        write_netcdf4(path, {'x': 2, 'y': 3},
                      [('x', ('y',), np.zeros(3), {})])
    with pytest.raises(ValueError, match='does not match'):
        write_netcdf4(path, {'x': 2}, [('v', ('x',), np.zeros(3), {})])


def test_dataset_to_netcdf4_and_back(tmp_path):
    """``write_netcdf(format='NETCDF4')`` -> ``open_dataset``: a Dataset with
    NaNs, a coordinate, text and a record dimension survives the round trip
    (the record dimension becomes a fixed one: contiguous storage)."""
    import pyremap_amd
    from pyremap_amd.io.netcdf import file_format, write_netcdf
    ds = pyremap_amd.Dataset(attrs={'title': 'round trip'})
    t = np.arange(24.0).reshape(2, 3, 4)
    t[0, 1, 2] = np.nan
    ds['lat'] = pyremap_amd.DataArray(np.array([-10.0, 0.0, 10.0]),
                                      dims=('lat',), attrs={'units': 'deg'})
    ds['t'] = pyremap_amd.DataArray(t, dims=('Time', 'lat', 'lon'),
                                    attrs={'units': 'K'})
    ds['xtime'] = pyremap_amd.DataArray(
        np.frombuffer(b'0001-01-010001-01-02', dtype='S1').reshape(2, 10),
        dims=('Time', 'StrLen'))
    path = str(tmp_path / 'ds4.nc')
    write_netcdf(ds, path, format='NETCDF4', unlimited_dims=['Time'])
    assert file_format(path) == 'NETCDF4'
    back = netcdf.open_dataset(path)
    assert back['t'].dims == ('Time', 'lat', 'lon')
    np.testing.assert_array_equal(back['t'].values, t)       # NaN restored
    assert back['t'].attrs['units'] == 'K'
    assert 'lat' in back.coords and back['lat'].attrs['units'] == 'deg'
    assert bytes(back['xtime'].values[1]) == b'0001-01-02'
    assert back.attrs['title'] == 'round trip'
    raw = netcdf.open_dataset(path, mask_and_scale=False)
    assert raw['t'].values[0, 1, 2] == 9.969209968386869e+36


def test_mapping_file_written_as_netcdf4(tmp_path, expected):
    """``write_mapping(format='NETCDF4')``: the layout ESMF --netcdf4 uses,
    read back by ``read_mapping`` unchanged."""
    m = mapfile.read_mapping(os.path.join(HERE, 'map_nc4.nc'))
    for fmt in ('NETCDF4', 'NETCDF3_64BIT_DATA', None):
        path = str(tmp_path / f'copy_{fmt}.nc')
        mapfile.write_mapping(path, m.n_a, m.n_b, m.src_grid_dims,
                              m.dst_grid_dims, m.row, m.col, m.S, m.frac_b,
                              attrs={'normalization': 'destarea'},
                              format=fmt)
        want = {'NETCDF4': 'NETCDF4', None: 'NETCDF3_64BIT'}.get(fmt, fmt)
        assert netcdf.file_format(path) == want
        back = mapfile.read_mapping(path)
        assert (back.n_a, back.n_b, back.n_s) == (m.n_a, m.n_b, m.n_s)
        for name in ('row', 'col', 'S', 'frac_b', 'src_grid_dims',
                     'dst_grid_dims'):
            np.testing.assert_array_equal(getattr(back, name),
                                          getattr(m, name))
    with pytest.raises(ValueError, match='unknown mapping-file format'):
        mapfile.write_mapping(str(tmp_path / 'x.nc'), m.n_a, m.n_b,
                              m.src_grid_dims, m.dst_grid_dims, m.row, m.col,
                              m.S, m.frac_b, format='GRIB')
