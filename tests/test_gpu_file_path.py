"""
GPU tests of the callers either side of the kernel: mapping file on disk ->
Remapper -> remap_numpy, and the file -> file path (`ncremap`/`remap_file`).
"""
import logging
import os

import numpy as np
import pytest

from helpers import assert_bitwise

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def setup(tmp_path_factory):
    assert torch.cuda.is_available()
    from pyremap_amd import (
        LatLonGridDescriptor,
        MpasCellMeshDescriptor,
        synthetic,
    )
    tmp = tmp_path_factory.mktemp('files')
    n_cells, nlat, nlon = 1500, 18, 36
    m = synthetic.conservative_map(n_cells, (nlat, nlon), 1, 6, seed=21)
    map_path = str(tmp / 'map_toy_to_10deg_esmfaave.nc')
    m.save(map_path)
    rng = np.random.default_rng(4)
    src = MpasCellMeshDescriptor(mesh_name='toy', lat=rng.random(n_cells),
                                 lon=rng.random(n_cells))
    dst = LatLonGridDescriptor.create(np.linspace(-90, 90, nlat + 1),
                                      np.linspace(-180, 180, nlon + 1))
    return dict(tmp=tmp, map=m, map_path=map_path, src=src, dst=dst,
                n_cells=n_cells, nlat=nlat, nlon=nlon)


def _input_dataset(setup, fmt, path):
    from pyremap_amd import DataArray, Dataset
    from pyremap_amd.io.netcdf import write_netcdf
    rng = np.random.default_rng(9)
    n = setup['n_cells']
    ds = Dataset(attrs={'history': 'made for a test', 'source': 'MPAS'})
    temp = rng.standard_normal((2, n, 5))
    temp[:, rng.random(n) < 0.3, 3:] = np.nan        # sea floor
    ds['temperature'] = DataArray(temp, dims=('Time', 'nCells',
                                              'nVertLevels'),
                                  attrs={'units': 'C'})
    ds['ssh'] = DataArray(rng.standard_normal((2, n)).astype(np.float32),
                          dims=('Time', 'nCells'))
    ds['daysSinceStart'] = DataArray(np.asarray([0.5, 1.5]), dims=('Time',))
    ds['xtime'] = DataArray(np.frombuffer(b'0001-01-01_00:00:00'
                                          b'0001-01-02_00:00:00',
                                          dtype='S1').reshape(2, 19),
                            dims=('Time', 'StrLen'))
    write_netcdf(ds, path, format=fmt, unlimited_dims=['Time'])
    return ds


def test_mapping_file_on_disk_equals_in_memory(setup):
    from oracle import oracle
    from pyremap_amd import Remapper
    m = setup['map']
    mm = m.numpy()
    r = Remapper(map_filename=setup['map_path'],
                 src_descriptor=setup['src'], dst_descriptor=setup['dst'])
    plan = r.load_mapping()
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    rowptr, col, val = plan.to_host_csr()
    assert np.array_equal(rowptr, csr.indptr)
    assert np.array_equal(col, csr.indices)
    assert_bitwise(val, csr.data)
    assert r._matrix is plan and r.load_mapping() is plan     # cached


@pytest.mark.parametrize('fmt', ['NETCDF3_64BIT', 'NETCDF3_64BIT_DATA'])
def test_ncremap_equals_remap_numpy(setup, fmt):
    """The contract of tests/test_interpolate.py:192-240 of the reference:
    the file path and remap_numpy give the same variables and values."""
    from oracle import oracle
    from pyremap_amd import Remapper
    from pyremap_amd.io.netcdf import open_dataset
    tmp = setup['tmp']
    src_path = str(tmp / f'in_{fmt}.nc')
    out_path = str(tmp / f'out_{fmt}.nc')
    ds = _input_dataset(setup, fmt, src_path)
    r = Remapper(map_filename=setup['map_path'],
                 src_descriptor=setup['src'], dst_descriptor=setup['dst'])
    log = logging.getLogger('remap_file_test')
    r.ncremap(src_path, out_path, renormalize=0.01, logger=log)
    out = open_dataset(out_path)
    ref = r.remap_numpy(open_dataset(src_path), renormalization_threshold=0.01)
    assert out.encoding['format'] == fmt
    assert out.encoding['unlimited_dims'] == ['Time']
    assert list(out.data_vars) == list(ref.data_vars) == [
        'temperature', 'ssh', 'daysSinceStart', 'xtime']
    assert sorted(out.coords) == ['lat', 'lon']
    assert out['temperature'].dims == ('Time', 'lat', 'lon', 'nVertLevels')
    assert out['ssh'].dims == ('Time', 'lat', 'lon')
    assert out['ssh'].dtype == np.float64         # always float64 out
    assert out.attrs['mesh_name'] == '10.0x10.0degree'
    assert out.attrs['history'].startswith('made for a test\n')
    for name in ('temperature', 'ssh', 'daysSinceStart'):
        assert_bitwise(out[name].values, ref[name].values, name)
    assert out['xtime'].values.tobytes() == ds['xtime'].values.tobytes()
    np.testing.assert_array_equal(out['lat'].values, setup['dst'].lat)
    # ... and both equal the oracle on the same triplets
    m = setup['map']
    mm = m.numpy()
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    temp = ds['temperature'].values
    expect = oracle.remap_numpy_array(
        csr, mm['frac_b'], m.dst_dims,
        np.ma.masked_array(temp, np.isnan(temp)), [1], 0.01).filled(np.nan)
    assert_bitwise(out['temperature'].values, expect)
    assert np.isnan(expect).any() and not np.isnan(expect).all()

    # overwrite=False + existing file: silent no-op (ncremap.py:18-19)
    before = os.path.getmtime(out_path)
    r.remap_file(src_path, out_path, renormalize=None)
    assert os.path.getmtime(out_path) == before
    # variable_list (ncremap.py:64-65)
    sub_path = str(tmp / f'sub_{fmt}.nc')
    r.ncremap(src_path, sub_path, variable_list=['ssh'], overwrite=True)
    sub = open_dataset(sub_path)
    assert list(sub.data_vars) == ['ssh']
    full = r.remap_numpy(open_dataset(src_path))['ssh'].values
    assert_bitwise(sub['ssh'].values, full)


def test_file_path_errors(setup):
    from pyremap_amd import PointCollectionDescriptor, Remapper
    r = Remapper(src_descriptor=setup['src'], dst_descriptor=setup['dst'],
                 map_filename=None)
    r.map_filename = None
    with pytest.raises(ValueError, match='No mapping file'):
        # _setup_remapper would invent a default name; the reference raises
        # from _validate_inputs only when the name is unset afterwards
        from pyremap_amd.remapper.remap_file import _remap_file
        _remap_file(r, 'a.nc', 'b.nc', None, False, None, None, False)
    pts = PointCollectionDescriptor(np.zeros(3), np.zeros(3), 'pts')
    r = Remapper(src_descriptor=pts, dst_descriptor=setup['dst'],
                 map_filename=setup['map_path'])
    with pytest.raises(TypeError, match='point collection'):
        r.ncremap('a.nc', str(setup['tmp'] / 'never.nc'))


def test_netcdf4_mapping_and_field_files(tmp_path):
    """NetCDF-4 on both sides of the kernel (what ESMF ``--netcdf4`` and
    netCDF4/xarray produce), read with the package's own HDF5 reader: the
    weights of ``golden/hdf5/map_nc4.nc`` applied to the real NetCDF-4 field
    file of the reference's tests, against the oracle."""
    from oracle import oracle
    from pyremap_amd import (
        LatLonGridDescriptor,
        MpasCellMeshDescriptor,
        Remapper,
    )
    from pyremap_amd.io import mapfile
    from pyremap_amd.io.netcdf import open_dataset
    here = os.path.join(os.path.dirname(__file__), 'golden', 'hdf5')
    # (1) the tiny NetCDF-4 mapping file: 3 x 4 -> 2 x 4 lat-lon
    path = os.path.join(here, 'map_nc4.nc')
    m = mapfile.read_mapping(path)
    src = LatLonGridDescriptor.create(np.linspace(-30, 30, 4),
                                      np.linspace(0, 40, 5))
    dst = LatLonGridDescriptor.create(np.linspace(-30, 30, 3),
                                      np.linspace(0, 40, 5))
    r = Remapper(map_filename=path, src_descriptor=src, dst_descriptor=dst)
    rng = np.random.default_rng(2)
    field = rng.standard_normal((5, 3, 4, 70))
    field[rng.random(field.shape) < 0.2] = np.nan
    from pyremap_amd import DataArray
    out = r.remap_numpy(DataArray(field, dims=('t', 'lat', 'lon', 'z')),
                        renormalization_threshold=0.1)
    csr = oracle.coo_to_csr(m.row - 1, m.col - 1, m.S, m.n_b, m.n_a)
    ref = oracle.remap_numpy_array(
        csr, m.frac_b, (2, 4),
        np.ma.masked_array(field, mask=np.isnan(field)), [1, 2], 0.1)
    assert_bitwise(np.asarray(out.values), np.ma.filled(ref, np.nan),
                   'netcdf4 map')
    # (2) a real NetCDF-4 field file as the source of a remap
    ds = open_dataset(os.path.join(here, 'nc4_ref_latlon_to_mpas_cell.nc'))
    n_cells = ds['SST'].shape[1]
    from pyremap_amd import synthetic
    mm = synthetic.conservative_map(n_cells, (10, 20), 1, 5, seed=8)
    map_path = str(tmp_path / 'map_cells_to_latlon.nc')
    mm.save(map_path)
    src = MpasCellMeshDescriptor(mesh_name='qu240', size=n_cells)
    dst = LatLonGridDescriptor.create(np.linspace(-90, 90, 11),
                                      np.linspace(-180, 180, 21))
    r2 = Remapper(map_filename=map_path, src_descriptor=src,
                  dst_descriptor=dst)
    out2 = r2.remap_numpy(ds)
    assert out2['SST'].dims == ('time', 'lat', 'lon')
    host = mm.numpy()
    csr2 = oracle.coo_to_csr(host['row'] - 1, host['col'] - 1, host['S'],
                             mm.n_b, mm.n_a)
    sst = np.asarray(ds['SST'].values)
    if np.isnan(sst).any():
        sst = np.ma.masked_array(sst, mask=np.isnan(sst))
    ref2 = oracle.remap_numpy_array(csr2, host['frac_b'], (10, 20), sst,
                                    [1], None)
    assert_bitwise(np.asarray(out2['SST'].values),
                   np.ma.filled(ref2, np.nan), 'netcdf4 field')
    assert np.array_equal(out2['date'].values, ds['date'].values)


def test_build_map_analytic_then_remap(tmp_path):
    """``Remapper(map_tool='analytic').build_map()`` writes the mapping file
    under the reference's default name and the GPU applies it: a conservative
    1 deg -> 2.5 deg remap keeps the area integral of the field; the result
    equals the oracle's bit for bit."""
    from oracle import oracle
    from pyremap_amd import DataArray, Remapper, get_lat_lon_descriptor
    from pyremap_amd.io import mapfile
    src = get_lat_lon_descriptor(1.0, 1.0)
    dst = get_lat_lon_descriptor(2.5, 2.5)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        r = Remapper(method='conserve', map_tool='analytic',
                     src_descriptor=src, dst_descriptor=dst)
        r.build_map()
        assert r.map_filename == \
            'map_1.0x1.0degree_to_2.5x2.5degree_analyticaave.nc'
        assert os.path.exists(r.map_filename)
        lat, lon = np.meshgrid(src.lat, src.lon, indexing='ij')
        rng = np.random.default_rng(3)
        field = np.cos(np.radians(lat))[None] * (
            1.0 + 0.3 * np.sin(np.radians(3 * lon))[None] +
            0.01 * rng.standard_normal((6,) + lat.shape))
        out = r.remap_numpy(DataArray(field, dims=('t', 'lat', 'lon')))
        m = mapfile.read_mapping(r.map_filename)
    finally:
        os.chdir(cwd)
    csr = oracle.coo_to_csr(m.row - 1, m.col - 1, m.S, m.n_b, m.n_a)
    ref = oracle.remap_numpy_array(csr, m.frac_b, (len(dst.lat),
                                                   len(dst.lon)),
                                   field, [1, 2], None)
    got = np.asarray(out.values)
    assert_bitwise(got, np.ma.filled(ref, np.nan), 'analytic conserve')

    def areas(d):
        dsin = np.diff(np.sin(np.radians(d.lat_corner)))
        return dsin[:, None] * np.radians(np.diff(d.lon_corner))[None, :]

    before = (field * areas(src)[None]).sum(axis=(1, 2))
    after = (got * areas(dst)[None]).sum(axis=(1, 2))
    assert np.allclose(after, before, rtol=1e-12)
    with pytest.raises(NotImplementedError, match='analytic'):
        Remapper(src_descriptor=src, dst_descriptor=dst).build_map()


def test_reference_fixture_latlon_file_to_latlon_array(tmp_path):
    """
    The reference's ``test_latlon_file_to_latlon_array``
    (tests/test_interpolate.py:492-516) replayed on its own data files: the
    1 deg SST file -> a 2 deg grid, through ``ncremap`` (file -> file) and
    ``remap_numpy``, compared with the output the reference stored.  The
    weights there came from ESMF, here from ``pyremap_amd.weights``
    (``bilinear_3d``: ESMF's construction, reproduced), and SST agrees at the
    reference's own rtol 1e-5 -- in fact to 1e-10 K; everything structural
    -- dims, coordinates, untouched variables, dtypes, attributes -- is exact.
    """
    from pyremap_amd import LatLonGridDescriptor, Remapper
    from pyremap_amd.io.netcdf import open_dataset
    here = os.path.join(os.path.dirname(__file__), 'golden', 'ref_fixtures')
    in_filename = os.path.join(here, 'SST_annual_1870-1900.nc')
    ref_filename = os.path.join(here, 'ref_latlon_file_to_latlon_array.nc')
    src = LatLonGridDescriptor.read(in_filename, lat_var_name='lat',
                                    lon_var_name='lon')
    assert src.mesh_name == '1.0x1.0degree' and src.regional is False
    assert src.units == 'degrees'
    dst = LatLonGridDescriptor.create(np.linspace(-90.0, 90.0, 91),
                                      np.linspace(-180.0, 180.0, 181),
                                      units='degrees')
    remapper = Remapper(ntasks=1, map_filename=str(tmp_path / 'map.nc'),
                        method='bilinear', map_tool='analytic',
                        use_tmp=False, src_descriptor=src,
                        dst_descriptor=dst)
    remapper.build_map()
    assert os.path.exists(remapper.map_filename)
    out_filename = str(tmp_path / 'out.nc')
    remapper.ncremap(in_filename=in_filename, out_filename=out_filename,
                     replace_mpas_fill=True)
    ds_file = open_dataset(out_filename)
    ds_mem = remapper.remap_numpy(open_dataset(in_filename), 0.01)
    ds_ref = open_dataset(ref_filename)
    for ds in (ds_file, ds_mem):
        # assertDimsEqual of the reference's tests/__init__.py
        for name in ds_ref.data_vars:
            assert name in ds.variables, name
            assert ds[name].dims == ds_ref[name].dims, name
            assert ds[name].shape == ds_ref[name].shape, name
        np.testing.assert_array_equal(ds['lat'].values, ds_ref['lat'].values)
        np.testing.assert_array_equal(ds['lon'].values, ds_ref['lon'].values)
        for name in ('date', 'datesec', 'date_frac'):
            np.testing.assert_array_equal(ds[name].values,
                                          ds_ref[name].values)
            assert ds[name].dtype == ds_ref[name].dtype
        sst, want = ds['SST'].values, ds_ref['SST'].values
        assert sst.dtype == np.float64 == want.dtype    # f32 in, f64 out
        assert np.array_equal(np.isnan(sst), np.isnan(want))
        # the reference's own tolerance (tests/__init__.py:62-67) -- and in
        # fact the rounding of the stored float64 file: the weights are
        # ESMF's (weights.bilinear_3d)
        assert np.isclose(sst, want, rtol=1e-5, atol=1e-8).all()
        assert np.abs(sst - want).max() < 1e-10
        assert ds['SST'].attrs['units'] == ds_ref['SST'].attrs['units']
        assert ds.attrs['title'] == ds_ref.attrs['title']
    # the two paths of this package agree with each other exactly
    assert_bitwise(ds_file['SST'].values, ds_mem['SST'].values,
                   'ncremap vs remap_numpy')


@pytest.mark.parametrize('kind', ['mpas_cell', 'mpas_cell_expand',
                                  'point_collection'])
def test_reference_fixture_latlon_to_points(tmp_path, kind):
    """
    The reference's ``test_latlon_to_mpas_cell`` and
    ``test_latlon_file_to_point_collection`` (tests/test_interpolate.py)
    (and ``test_latlon_to_mpas_cell_expand``: the same map built with
    ``expand_dist`` / ``expand_factor`` set -- they widen the SCRIP corners
    of the destination cells, which bilinear does not read)
    replayed: the 1 deg SST file -> the 7153 QU240 cell centres (taken from
    the stored outputs, which carry them as coordinates), bilinear.  SST
    agrees with what the reference stored (ESMF weights) at the reference's
    own tolerance (to the rounding of the stored files); dims, coordinates
    and pass-through variables exactly.
    """
    from pyremap_amd import (
        LatLonGridDescriptor,
        MpasCellMeshDescriptor,
        PointCollectionDescriptor,
        Remapper,
    )
    from pyremap_amd.io.netcdf import open_dataset
    gold = os.path.join(os.path.dirname(__file__), 'golden')
    in_filename = os.path.join(gold, 'ref_fixtures',
                               'SST_annual_1870-1900.nc')
    if kind == 'mpas_cell':
        ds_ref = open_dataset(os.path.join(
            gold, 'hdf5', 'nc4_ref_latlon_to_mpas_cell.nc'))
        dst = MpasCellMeshDescriptor(mesh_name='oQU240',
                                     lat=ds_ref['lat_cell'].values,
                                     lon=ds_ref['lon_cell'].values)
        dim, coords = 'nCells', ('lat_cell', 'lon_cell')
    elif kind == 'mpas_cell_expand':
        ds_ref = open_dataset(os.path.join(
            gold, 'ref_fixtures', 'ref_latlon_to_mpas_cell_expand.nc'))
        dst = MpasCellMeshDescriptor(
            os.path.join(gold, 'ref_fixtures', 'mpasMesh.nc'),
            mesh_name='oQU240')
        dim, coords = 'nCells', ()
    else:
        ds_ref = open_dataset(os.path.join(
            gold, 'ref_fixtures', 'ref_latlon_file_to_point_collection.nc'))
        dst = PointCollectionDescriptor(ds_ref['lat'].values,
                                        ds_ref['lon'].values,
                                        collection_name='mpas_cells',
                                        units='degrees',
                                        out_dimension='n_points')
        dim, coords = 'n_points', ('lat', 'lon')
    src = LatLonGridDescriptor.read(in_filename)
    remapper = Remapper(map_filename=str(tmp_path / f'map_{kind}.nc'),
                        method='bilinear', map_tool='analytic',
                        src_descriptor=src, dst_descriptor=dst)
    if kind == 'mpas_cell_expand':
        remapper.expand_dist = 1e5
        remapper.expand_factor = 1.2
    remapper.build_map()
    out = remapper.remap_numpy(open_dataset(in_filename), 0.01)
    assert out['SST'].dims == ('time', dim) == ds_ref['SST'].dims
    assert out['SST'].shape == ds_ref['SST'].shape == (1, 7153)
    assert out['SST'].dtype == np.float64
    for c in coords:
        np.testing.assert_array_equal(out[c].values, ds_ref[c].values)
        assert out[c].dims == (dim,)
    want = np.asarray(ds_ref['SST'].values, dtype=np.float64)
    got = out['SST'].values
    assert not np.isnan(got).any()
    # the reference's own tolerance; the stored files hold float32 (NCO) or
    # float64 values: equal to THEIR rounding, the cell beyond the last
    # latitude row (ESMF's pole cap) included
    assert np.isclose(got, want, rtol=1e-5, atol=1e-8).all()
    assert np.abs(got - want).max() < (1e-10 if kind == 'mpas_cell'
                                       else 2e-6)
    for name in ('date', 'datesec', 'date_frac'):
        np.testing.assert_array_equal(out[name].values, ds_ref[name].values)
    # file -> file gives the same numbers
    out_filename = str(tmp_path / f'out_{kind}.nc')
    remapper.ncremap(in_filename, out_filename)
    assert_bitwise(open_dataset(out_filename)['SST'].values, got,
                   'ncremap vs remap_numpy')
    # a point collection cannot be the SOURCE of ncremap (ncremap.py:20-23)
    if kind == 'point_collection':
        back = Remapper(map_filename=remapper.map_filename,
                        src_descriptor=dst, dst_descriptor=src)
        with pytest.raises(TypeError, match='point collection'):
            back.ncremap(in_filename, str(tmp_path / 'never.nc'))


def test_reference_fixture_stereographic_to_latlon():
    """
    The reference's ``test_stereographic_array_to_latlon_array``
    (tests/test_interpolate.py:576-621) replayed: the latitude of a 100 km
    Antarctic stereographic grid, as a ``(dim0, y, x, dim3)`` field, remapped
    to a 2 deg lat-lon grid (a two-axis source in the middle of a 4-D array).
    The stored reference output pins (i) this package's polar stereographic
    projection against pyproj, (ii) which destination cells are mapped at all
    -- the NaN pattern is IDENTICAL, 13 720 of 16 200 cells -- and (iii) the
    values at the reference's own tolerance (they agree to 6e-9 degrees: the
    weights are ESMF's).
    """
    from pyremap_amd import (
        DataArray,
        Dataset,
        LatLonGridDescriptor,
        ProjectionGridDescriptor,
        Remapper,
    )
    from pyremap_amd.io.netcdf import open_dataset
    from pyremap_amd.polar import get_antarctic_stereographic_projection
    here = os.path.join(os.path.dirname(__file__), 'golden', 'ref_fixtures')
    ds_ref = open_dataset(os.path.join(here, 'ref_stereographic_to_latlon.nc'))
    x_max, y_max, res = 3000e3, 2500e3, 100e3
    nx = 2 * int(x_max / res) + 1
    ny = 2 * int(y_max / res) + 1
    src = ProjectionGridDescriptor.create(
        get_antarctic_stereographic_projection(),
        np.linspace(-x_max, x_max, nx), np.linspace(-y_max, y_max, ny),
        f'{int(res * 1e-3)}km_Antarctic_stereo')
    dst = LatLonGridDescriptor.create(np.linspace(-90.0, 90.0, 91),
                                      np.linspace(-180.0, 180.0, 181),
                                      units='degrees')
    lat2d = src.coords['lat']['data']
    in_field = np.reshape(lat2d, (1, ny, nx, 1)).repeat(3, axis=0).repeat(
        2, axis=3)
    ds = Dataset()
    ds['complicated'] = DataArray(in_field, dims=('dim0', 'y', 'x', 'dim3'))
    remapper = Remapper(method='bilinear', map_tool='analytic',
                        map_filename='<memory>', src_descriptor=src,
                        dst_descriptor=dst)
    from pyremap_amd.weights import build_weights
    m = build_weights(src, dst, 'bilinear')
    remapper = Remapper.from_triplets(m.row, m.col, m.S, m.frac_b, src, dst)
    out = remapper.remap_numpy(ds, 0.01)
    got = out['complicated'].values
    want = ds_ref['complicated'].values
    assert out['complicated'].dims == ds_ref['complicated'].dims == \
        ('dim0', 'lat', 'lon', 'dim3')
    assert got.shape == want.shape == (3, 90, 180, 2)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isnan(want).sum() == 13720 * 6
    ok = ~np.isnan(want)
    assert np.isclose(got[ok], want[ok], rtol=1e-5, atol=1e-8).all()
    assert np.abs(got - want)[ok].max() < 1e-7
    np.testing.assert_array_equal(out['lat'].values, ds_ref['lat'].values)
    np.testing.assert_array_equal(out['lon'].values, ds_ref['lon'].values)
    # all dim0 slices and both dim3 slices are the same remap
    assert_bitwise(got[0], got[2], 'dim0 slices')
    assert_bitwise(got[..., 0], got[..., 1], 'dim3 slices')


def test_reference_fixture_latlon_to_stereographic(tmp_path):
    """
    The reference's ``test_latlon_to_stereographic``
    (tests/test_interpolate.py:547-574) replayed: the SST file -> the
    stereographic grid.  The stored output also carries pyproj's latitudes
    and longitudes of that grid: ours agree to 1e-11 degrees.
    """
    from pyremap_amd import LatLonGridDescriptor, Remapper
    from pyremap_amd import get_polar_descriptor
    from pyremap_amd.io.netcdf import open_dataset
    here = os.path.join(os.path.dirname(__file__), 'golden', 'ref_fixtures')
    in_filename = os.path.join(here, 'SST_annual_1870-1900.nc')
    ds_ref = open_dataset(os.path.join(here, 'ref_latlon_to_stereographic.nc'))
    src = LatLonGridDescriptor.read(in_filename)
    dst = get_polar_descriptor(6000.0, 5000.0, 100.0, 100.0)
    assert np.abs(dst.coords['lat']['data'] -
                  ds_ref['lat'].values).max() < 1e-11
    assert np.abs(dst.coords['lon']['data'] -
                  ds_ref['lon'].values).max() < 1e-11
    remapper = Remapper(map_filename=str(tmp_path / 'map.nc'),
                        method='bilinear', map_tool='analytic',
                        src_descriptor=src, dst_descriptor=dst)
    remapper.build_map()
    out_filename = str(tmp_path / 'out.nc')
    remapper.ncremap(in_filename, out_filename)
    ds_file = open_dataset(out_filename)
    ds_mem = remapper.remap_numpy(open_dataset(in_filename), 0.01)
    want = np.asarray(ds_ref['SST'].values, dtype=np.float64)
    for ds in (ds_file, ds_mem):
        assert ds['SST'].dims == ('time', 'y', 'x') == ds_ref['SST'].dims
        got = ds['SST'].values
        assert got.shape == want.shape == (1, 51, 61)
        assert not np.isnan(got).any()
        assert np.isclose(got, want, rtol=1e-5, atol=1e-8).all()
        assert np.abs(got - want).max() < 2e-6      # float32 in the file
        for name in ('date', 'datesec', 'date_frac'):
            np.testing.assert_array_equal(ds[name].values,
                                          ds_ref[name].values)
        np.testing.assert_array_equal(ds['x'].values, dst.x)
    assert_bitwise(ds_file['SST'].values, ds_mem['SST'].values,
                   'ncremap vs remap_numpy')


# the variables NCO adds and the reference's tests drop before comparing
# (tests/test_interpolate.py:200-208)
NCO_EXTRAS = ('lat_bnds', 'lon_bnds', 'gw', 'area', 'nvertices',
              'lat_vertices', 'lon_vertices')


def _assert_as_the_reference_does(ds, ds_ref, what):
    """`assertDimsEqual` + `assertDatasetApproxEqual` of the reference's
    tests/__init__.py:59-99 at ITS tolerances (rtol 1e-5, atol 1e-8) -- and,
    stricter than the reference (which lets a NaN on either side pass), the
    same cells masked."""
    # (2-D lat / lon of a projection grid are coordinates to xarray -- the
    # file names them in a `coordinates` attribute -- and to this package's
    # result; the plain reader used for the stored file lists them as data)
    ref_vars = [v for v in ds_ref.data_vars
                if v not in NCO_EXTRAS and v not in ds.coords]
    assert set(ds.data_vars) == set(ref_vars), what
    for name in ds.coords:
        if name in ds_ref.variables and \
                ds_ref[name].values.dtype.kind == 'f':
            assert np.allclose(ds[name].values, ds_ref[name].values,
                               rtol=1e-5, atol=1e-8), (what, name)
    for name in ref_vars:
        assert set(ds[name].dims) == set(ds_ref[name].dims), (what, name)
        got, want = ds[name].values, ds_ref[name].values
        if got.dtype.kind in 'SU' or want.dtype.kind in 'SU':
            continue
        assert got.shape == want.shape, (what, name)
        assert np.array_equal(np.isnan(got), np.isnan(want)), (what, name)
        ok = ~np.isnan(want)
        assert np.isclose(got[ok], want[ok], rtol=1e-5, atol=1e-8).all(), \
            (what, name, np.abs(got[ok] - want[ok]).max())
        # (in fact they agree to rounding: the weights ARE ESMF's)
        assert np.isclose(got[ok], want[ok], rtol=1e-9, atol=0.0).all(), \
            (what, name)


@pytest.mark.parametrize('case', ['mpas_cell_to_latlon',
                                  'mpas_edge_to_latlon',
                                  'mpas_vertex_to_latlon',
                                  'mpas_cell_to_stereographic'])
def test_reference_fixture_mpas_source(tmp_path, case):
    """
    The reference's ``test_mpas_cell_to_latlon``, ``test_mpas_edge_to_latlon``,
    ``test_mpas_vertex_to_latlon`` and ``test_mpas_cell_to_stereographic``
    (tests/test_interpolate.py:418-545) replayed on ITS data files -- the real
    QU240 mesh with its real cell numbering, one month of real MPAS-Ocean
    output -- and checked against the outputs it stored, at its own
    tolerances: ``build_map`` (bilinear weights on the dual mesh, as ESMF
    makes them) -> ``ncremap(replace_mpas_fill=True)`` and
    ``remap_numpy(ds, 0.01)`` on the GPU.
    """
    from pyremap_amd import (
        LatLonGridDescriptor,
        MpasCellMeshDescriptor,
        MpasEdgeMeshDescriptor,
        MpasVertexMeshDescriptor,
        Remapper,
        get_polar_descriptor,
    )
    from pyremap_amd.io.netcdf import open_dataset
    gold = os.path.join(os.path.dirname(__file__), 'golden')
    here = os.path.join(gold, 'ref_fixtures')
    mesh = os.path.join(here, 'mpasMesh.nc')
    kind = case.split('_')[1]
    src = {'cell': MpasCellMeshDescriptor, 'edge': MpasEdgeMeshDescriptor,
           'vertex': MpasVertexMeshDescriptor}[kind](mesh, mesh_name='oQU240')
    in_filename = {
        'cell': os.path.join(here, 'timeSeries.0002-01-01.nc'),
        'edge': os.path.join(here, 'mpasAreaEdge.nc'),
        'vertex': os.path.join(gold, 'hdf5', 'nc4_mpasAreaVertex.nc')}[kind]
    if case.endswith('latlon'):
        dst = LatLonGridDescriptor.read(
            os.path.join(here, 'SST_annual_1870-1900.nc'),
            lat_var_name='lat', lon_var_name='lon')
    else:
        dst = get_polar_descriptor(6000.0, 5000.0, 100.0, 100.0)
    remapper = Remapper(ntasks=1, map_filename=str(tmp_path / 'weights.nc'),
                        method='bilinear', map_tool='analytic',
                        use_tmp=False, src_descriptor=src,
                        dst_descriptor=dst)
    remapper.build_map()
    assert os.path.exists(remapper.map_filename)
    ds_ref = open_dataset(os.path.join(here, f'ref_{case}.nc'))
    out_filename = str(tmp_path / 'remapped.nc')
    remapper.ncremap(in_filename=in_filename, out_filename=out_filename,
                     replace_mpas_fill=True)
    ds_file = open_dataset(out_filename)
    _assert_as_the_reference_does(ds_file, ds_ref, 'ncremap')
    ds_mem = remapper.remap_numpy(open_dataset(in_filename), 0.01)
    _assert_as_the_reference_does(ds_mem, ds_ref, 'remap_numpy')
    for name in ds_mem.data_vars:
        if ds_mem[name].dtype.kind == 'f':
            assert_bitwise(ds_file[name].values, ds_mem[name].values,
                           f'ncremap vs remap_numpy: {name}')
    # the land pattern is there (40 % of a global grid, none dropped)
    some = next(n for n in ds_mem.data_vars if ds_mem[n].dtype.kind == 'f')
    assert 0.2 < np.isnan(ds_mem[some].values).mean() < 0.7


def test_examples_run(tmp_path, monkeypatch):
    """examples/: the stereographic -> stereographic script of the reference's
    examples directory (analytic weights) and the apply-a-mapping-file
    script, end to end on generated files."""
    import importlib.util
    from pyremap_amd import DataArray, Dataset, synthetic
    from pyremap_amd.io.netcdf import open_dataset, write_netcdf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(
            name, os.path.join(root, 'examples', f'{name}.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    monkeypatch.chdir(tmp_path)
    # -- stereographic 50 km -> 20 km ---------------------------------------
    x = np.linspace(-1.0e6, 1.0e6, 41)
    y = np.linspace(-5.0e5, 5.0e5, 21)
    xx, yy = np.meshgrid(x, y)
    ds = Dataset()
    ds['x'] = DataArray(x, dims=('x',))
    ds['y'] = DataArray(y, dims=('y',))
    thick = 1000.0 + 1e-3 * xx - 2e-3 * yy
    ds['thickness'] = DataArray(np.stack([thick, 2 * thick]),
                                dims=('time', 'y', 'x'))
    write_netcdf(ds, 'stereo_in.nc')
    r = load('remap_stereographic').main(
        ['-i', 'stereo_in.nc', '-o', 'stereo_out.nc', '-r', '20'])
    assert os.path.exists(r.map_filename)
    out = open_dataset('stereo_out.nc')
    assert out['thickness'].shape == (2, 51, 101)
    xo, yo = np.meshgrid(out['x'].values, out['y'].values)
    # linear in x and y: reproduced to the curvature of a 50 km quad (the
    # quads live on the sphere, as ESMF's do, not in the projection plane)
    want = 1000.0 + 1e-3 * xo - 2e-3 * yo
    assert np.abs(out['thickness'].values[0] - want).max() < 0.02
    assert np.abs(out['thickness'].values[1] - 2 * want).max() < 0.04
    # -- apply a mapping file to an "MPAS" file -----------------------------
    m = synthetic.conservative_map(800, (18, 36), 1, 5, seed=5)
    m.save('map_toy_to_10x10degree_aave.nc')
    ds2 = Dataset()
    field = np.random.default_rng(0).standard_normal((3, 800, 4))
    field[:, :50, 2:] = -9.99999979021476795361e+33       # MPAS fill value
    ds2['temperature'] = DataArray(field, dims=('Time', 'nCells',
                                                'nVertLevels'))
    ds2['other'] = DataArray(field[..., 0], dims=('Time', 'nCells'))
    write_netcdf(ds2, 'mpas_in.nc')
    load('remap_with_mapping_file').main(
        ['-m', 'map_toy_to_10x10degree_aave.nc', '-i', 'mpas_in.nc', '-o',
         'mpas_out.nc', '--dlon', '10', '--dlat', '10', '-v', 'temperature',
         '--renormalize', '0.01'])
    out2 = open_dataset('mpas_out.nc')
    assert out2['temperature'].dims == ('Time', 'lat', 'lon', 'nVertLevels')
    assert out2['temperature'].shape == (3, 18, 36, 4)
    assert 'other' not in out2.variables
    assert np.nanmax(np.abs(out2['temperature'].values)) < 10.0   # no fills
    # -- the reference's make_mpas_to_lat_lon_mapping.py workflow, no ESMF ---
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                        'ref_fixtures')
    r3 = load('make_mpas_to_lat_lon_mapping').main(
        ['--mesh', os.path.join(gold, 'mpasMesh.nc'), '--mesh-name',
         'oQU240', '-i', os.path.join(gold, 'timeSeries.0002-01-01.nc'),
         '--res', '1.0', '-o', 'mpas_example'])
    assert os.path.basename(r3.map_filename) == \
        'map_oQU240_to_1.0x1.0degree_analyticbilin.nc'
    a = open_dataset('mpas_example/remapped_1.0x1.0degree_file.nc')
    b = open_dataset('mpas_example/remapped_1.0x1.0degree_array.nc')
    ref = open_dataset(os.path.join(gold, 'ref_mpas_cell_to_latlon.nc'))
    for name in ('timeMonthly_avg_ssh', 'timeMonthly_avg_tThreshMLD'):
        assert a[name].dims == ('Time', 'lat', 'lon')
        assert np.array_equal(a[name].values, b[name].values, equal_nan=True)
        # get_lat_lon_descriptor(1, 1) holds the cells of the reference's
        # stored output, longitudes from -179.5 instead of 0.5: the example
        # lands on it
        got = np.roll(a[name].values, 180, axis=-1)
        want = ref[name].values
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.isclose(got[ok], want[ok], rtol=1e-9).all()


# ---------------------------------------------------------------------------
# streaming: variable by variable with bounded host memory
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('fmt', ['NETCDF3_64BIT_DATA', 'NETCDF4'])
def test_ncremap_streamed_equals_eager(setup, fmt, monkeypatch):
    """
    With every variable above the streaming threshold the file path reads,
    remaps and writes one variable at a time (NCO's way,
    ncremap.py:117-145); the output is the one the all-at-once path writes:
    same variables, dims, attributes (`_FillValue` where NaNs turned up) and
    values, bit for bit.
    """
    from pyremap_amd import Remapper
    from pyremap_amd.io.netcdf import open_dataset
    from pyremap_amd.remapper import remap_file
    tmp = setup['tmp']
    src_path = str(tmp / f'stream_in_{fmt}.nc')
    _input_dataset(setup, fmt, src_path)
    outs = {}
    for tag, threshold in (('eager', 1 << 40), ('streamed', 1)):
        monkeypatch.setattr(remap_file, 'STREAM_BYTES', threshold)
        r = Remapper(map_filename=setup['map_path'],
                     src_descriptor=setup['src'],
                     dst_descriptor=setup['dst'])
        out_path = str(tmp / f'stream_out_{tag}_{fmt}.nc')
        r.ncremap(src_path, out_path, renormalize=0.01, overwrite=True)
        outs[tag] = open_dataset(out_path, mask_and_scale=False)
    a, b = outs['eager'], outs['streamed']
    assert list(a.data_vars) == list(b.data_vars)
    assert sorted(a.coords) == sorted(b.coords)
    assert {k: str(v) for k, v in a.attrs.items()} == \
        {k: str(v) for k, v in b.attrs.items()}
    for name in a.variables:
        va, vb = a.variables[name], b.variables[name]
        assert va.dims == vb.dims and va.dtype == vb.dtype, name
        assert sorted(va.attrs) == sorted(vb.attrs), name
        assert va.values.tobytes() == vb.values.tobytes(), name
    assert '_FillValue' in b.variables['temperature'].attrs
    assert '_FillValue' not in b.variables['daysSinceStart'].attrs


_STREAM_SCRIPT = r'''
import os, resource, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from pyremap_amd import (DataArray, Dataset, LatLonGridDescriptor,
                         MpasCellMeshDescriptor, Remapper, synthetic)
from pyremap_amd.io.netcdf import write_netcdf
tmp, mode = sys.argv[2], sys.argv[3]
n, nlat, nlon, T, L = 60000, 200, 300, 8, 24
map_path = os.path.join(tmp, 'map.nc')
src_path = os.path.join(tmp, 'big_in.nc')
rng = np.random.default_rng(1)
if not os.path.exists(src_path):
    synthetic.conservative_map(n, (nlat, nlon), 2, 6, seed=3,
                               locality='mesh').save(map_path)
    ds = Dataset()
    for v in range(4):
        x = rng.standard_normal((T, n, L))
        if v % 2:
            x[:, rng.random(n) < 0.2, L // 2:] = np.nan
        ds[f'var{v}'] = DataArray(x, dims=('Time', 'nCells', 'nVertLevels'))
    write_netcdf(ds, src_path, format='NETCDF3_64BIT_DATA',
                 unlimited_dims=['Time'])
    del ds, x
if mode == 'make':
    # (its own process: the peak of BUILDING the file must not hide the
    # growth measured below)
    sys.exit(0)
src = MpasCellMeshDescriptor(mesh_name='toy', lat=rng.random(n),
                             lon=rng.random(n))
dst = LatLonGridDescriptor.create(np.linspace(-90, 90, nlat + 1),
                                  np.linspace(-180, 180, nlon + 1))
r = Remapper(map_filename=map_path, src_descriptor=src, dst_descriptor=dst)
r.load_mapping()
r.remap_array(np.zeros((n, 40)), [0], 0.1)        # runtime + kernels loaded
# growth of the resident set ACROSS the call: the high-water mark of the
# process (ru_maxrss) misses it whenever start-up -- loading the runtime and
# the code objects -- peaked higher than the steady state the call starts
# from, so the current size is sampled as well and the larger growth counts
import threading
page = os.sysconf('SC_PAGE_SIZE')
def rss():
    with open('/proc/self/statm') as f:
        return int(f.read().split()[1]) * page
seen = [rss()]
done = threading.Event()
def sampler():
    while not done.is_set():
        seen[0] = max(seen[0], rss())
        time.sleep(0.001)
base_now = rss()
base = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
th = threading.Thread(target=sampler, daemon=True)
th.start()
t0 = time.perf_counter()
r.ncremap(src_path, os.path.join(tmp, f'big_out_{mode}.nc'), renormalize=0.05,
          overwrite=True)
dt = time.perf_counter() - t0
seen[0] = max(seen[0], rss())
done.set()
th.join()
peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
grown = max((peak - base) * 1024, seen[0] - base_now)
print('RESULT', mode, grown, dt, base * 1024)
'''


def test_ncremap_streams_with_bounded_memory(tmp_path):
    """
    A file of four 92 MB variables (output 4 x 92 MB): the all-at-once path
    holds every input and every result until the write; the streaming path
    holds about two variables.  Measured as the growth of the process's peak
    RSS across the call, in fresh processes; both write the same bytes.
    """
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'stream_probe.py'
    script.write_text(_STREAM_SCRIPT)
    grown, secs = {}, {}
    for mode, threshold in (('make', '1'), ('eager', str(1 << 40)),
                            ('streamed', '1')):
        env = dict(os.environ, PYREMAP_AMD_STREAM_BYTES=threshold)
        proc = subprocess.run(
            [sys.executable, str(script), repo, str(tmp_path), mode],
            capture_output=True, text=True, env=env, timeout=900)
        assert proc.returncode == 0, proc.stderr[-3000:]
        if mode == 'make':
            continue
        line = [ln for ln in proc.stdout.splitlines()
                if ln.startswith('RESULT')][-1].split()
        grown[mode], secs[mode] = int(line[2]), float(line[3])
    one_in = 8 * 60000 * 24 * 8
    one_out = 8 * 200 * 300 * 24 * 8
    # streamed: two variables in flight, whatever the file holds -- where
    # all at once holds the four inputs (and, when its results land in
    # pageable memory, the four results: 3-4 x more; with pinned results,
    # which stay on the device until the writer asks, about as much again as
    # one result.  tools/stream_timing.py has the 8 GB file: + 2.2 GB against
    # + 10.8 GB)
    assert grown['streamed'] < 2.6 * (one_in + one_out), (grown, secs)
    assert grown['streamed'] < grown['eager'] + (32 << 20), (grown, secs)
    assert grown['eager'] > 2 * one_in, (grown, secs)
    assert secs['streamed'] < 2.0 * secs['eager'] + 0.5, secs
    a = open(tmp_path / 'big_out_eager.nc', 'rb').read()
    b = open(tmp_path / 'big_out_streamed.nc', 'rb').read()
    # same data bytes; the streamed header may be followed by free space
    from pyremap_amd.io.netcdf import open_dataset
    da = open_dataset(str(tmp_path / 'big_out_eager.nc'),
                      mask_and_scale=False)
    db = open_dataset(str(tmp_path / 'big_out_streamed.nc'),
                      mask_and_scale=False)
    assert abs(len(a) - len(b)) < 4096
    for name in da.variables:
        assert da.variables[name].values.tobytes() == \
            db.variables[name].values.tobytes(), name
        assert sorted(da.variables[name].attrs) == \
            sorted(db.variables[name].attrs), name


def test_mapping_files_are_loaded_once_per_process(setup, tmp_path):
    """Short-lived Remappers over one mapping file (MPAS-Analysis builds one
    per variable group) find the device plan of the first; a rewritten file
    is loaded afresh; descriptors are validated per Remapper all the same."""
    import shutil
    import time

    from pyremap_amd import LatLonGridDescriptor, Remapper
    from pyremap_amd.remapper import remap_numpy
    path = str(tmp_path / 'map_cached.nc')
    shutil.copy(setup['map_path'], path)

    def make(dst=None):
        return Remapper(map_filename=path, src_descriptor=setup['src'],
                        dst_descriptor=dst or setup['dst'])
    a, b = make(), make()
    plan = a.load_mapping()
    assert b.load_mapping() is plan and b.schedule == a.schedule
    x = np.random.default_rng(0).standard_normal((setup['n_cells'], 40))
    assert_bitwise(np.ma.filled(b.remap_array(x, [0]), np.nan),
                   np.ma.filled(a.remap_array(x, [0]), np.nan))
    # a descriptor that does not fit the file still raises
    wrong = LatLonGridDescriptor.create(np.linspace(-90, 90, 10),
                                        np.linspace(-180, 180, 37))
    with pytest.raises(ValueError, match="don't have the same size"):
        make(wrong).load_mapping()
    # the file changes: a new plan
    time.sleep(0.01)
    m2 = setup['map']
    m2.save(path)
    os.utime(path, ns=(time.time_ns(), time.time_ns()))
    assert make().load_mapping() is not plan
    # PYREMAP_AMD_PLAN_CACHE bounds the cache
    assert len(remap_numpy._PLAN_CACHE) <= remap_numpy._PLAN_CACHE_SIZE
