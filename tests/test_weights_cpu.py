"""
Analytic weight generation against brute-force
quadrature and the invariants a conservative / bilinear map must satisfy.
CPU only; the weights are applied on the GPU in tests/test_gpu_file_path.py.
"""
import os

import numpy as np
import pytest

from pyremap_amd import (
    LatLonGridDescriptor,
    get_lat_lon_descriptor,
    get_polar_descriptor,
)
from pyremap_amd.io import mapfile
from pyremap_amd.weights import (
    build_weights,
    overlap_1d,
    write_weights,
)


def _dense(m):
    A = np.zeros((m.n_b, m.n_a))
    np.add.at(A, (m.row - 1, m.col - 1), m.S)
    return A


def _areas(d):
    scale = np.pi / 180.0 if 'deg' in d.units else 1.0
    dlon = np.abs(np.diff(d.lon_corner)) * scale
    dsin = np.abs(np.diff(np.sin(np.clip(d.lat_corner * scale,
                                         -np.pi / 2, np.pi / 2))))
    return (dsin[:, None] * dlon[None, :]).reshape(-1)


def test_overlap_1d_against_brute_force():
    rng = np.random.default_rng(0)
    for trial in range(20):
        se = np.sort(rng.uniform(0, 10, rng.integers(2, 12)))
        de = np.sort(rng.uniform(-2, 12, rng.integers(2, 12)))
        if trial % 3 == 0:
            se = se[::-1].copy()
        if trial % 4 == 0:
            de = de[::-1].copy()
        j, i, length = overlap_1d(se, de)
        got = np.zeros((len(de) - 1, len(se) - 1))
        got[j, i] = length
        for a in range(len(de) - 1):
            for b in range(len(se) - 1):
                lo = max(min(de[a], de[a + 1]), min(se[b], se[b + 1]))
                hi = min(max(de[a], de[a + 1]), max(se[b], se[b + 1]))
                assert abs(got[a, b] - max(hi - lo, 0.0)) < 1e-14


def test_overlap_1d_periodic():
    se = np.linspace(-180.0, 180.0, 13)          # 30 degree cells
    de = np.linspace(0.0, 360.0, 9)              # 45 degree cells, shifted
    j, i, length = overlap_1d(se, de, period=360.0)
    got = np.zeros((8, 12))
    np.add.at(got, (j, i), length)
    assert np.allclose(got.sum(axis=1), 45.0)    # every dst cell covered
    assert np.allclose(got.sum(axis=0), 30.0)    # every src cell used up


@pytest.mark.parametrize('src_res, dst_res', [((10.0, 5.0), (4.0, 3.0)),
                                              ((3.0, 2.0), (7.5, 6.0))])
def test_conservative_global_lat_lon(src_res, dst_res):
    src = get_lat_lon_descriptor(*src_res)
    dst = get_lat_lon_descriptor(*dst_res, lon_min=0.0, lon_max=360.0)
    m = build_weights(src, dst, 'conserve')
    A = _dense(m)
    area_a, area_b = _areas(src), _areas(dst)
    assert np.allclose(m.frac_b, 1.0, atol=1e-12)
    assert np.allclose(A.sum(axis=1), 1.0, atol=1e-12)     # constants kept
    # conservation: every source cell's area is handed out exactly once
    assert np.allclose(area_b @ A, area_a, rtol=1e-12, atol=1e-15)
    # against quadrature of one destination cell
    scale = np.pi / 180.0
    row = (len(dst.lat) // 3) * len(dst.lon) + 5
    jy, jx = divmod(row, len(dst.lon))
    n = 400
    lat = np.linspace(dst.lat_corner[jy], dst.lat_corner[jy + 1], n + 1)
    lon = np.linspace(dst.lon_corner[jx], dst.lon_corner[jx + 1], n + 1)
    latc = 0.5 * (lat[:-1] + lat[1:])
    lonc = (0.5 * (lon[:-1] + lon[1:]) + 180.0) % 360.0 - 180.0
    w = np.diff(np.sin(lat * scale))[:, None] * np.ones(n)[None, :]
    iy = np.searchsorted(src.lat_corner, latc) - 1
    ix = np.searchsorted(src.lon_corner, lonc) - 1
    quad = np.zeros(m.n_a)
    np.add.at(quad, (iy[:, None] * len(src.lon) + ix[None, :]).reshape(-1),
              w.reshape(-1))
    quad /= quad.sum()
    assert np.abs(A[row] - quad).max() < 2e-3


def test_conservative_regional_source_gives_frac_b():
    src = LatLonGridDescriptor.create(np.linspace(-30, 30, 13),
                                      np.linspace(10, 100, 19))
    dst = get_lat_lon_descriptor(20.0, 20.0)
    assert src.regional and not dst.regional
    m = build_weights(src, dst, 'conserve')
    A = _dense(m)
    assert np.allclose(A.sum(axis=1), m.frac_b)
    fb = m.frac_b.reshape(len(dst.lat), len(dst.lon))
    assert fb.max() <= 1.0 and fb.min() == 0.0
    assert np.isclose(fb[4, 10], 1.0)          # 0..20 N, 20..40 E: inside
    assert 0.0 < fb[4, 9] < 1.0                # 0..20 E: half covered
    assert np.allclose(_areas(dst) @ A, _areas(src), rtol=1e-12)


def test_bilinear_lat_lon_is_esmfs_bilinear():
    """Quads of four source centres joined by straight 3-D lines: the
    weighted corner sum points AT the destination point; a global source is
    capped at the poles by a node standing for the mean of the last row."""
    from pyremap_amd.weights import _unit
    src = get_lat_lon_descriptor(10.0, 10.0)
    dst = get_lat_lon_descriptor(2.5, 2.5)
    m = build_weights(src, dst, 'bilinear')
    A = _dense(m)
    assert np.all(m.frac_b == 1.0)
    assert np.allclose(A.sum(axis=1), 1.0, rtol=0, atol=1e-14)
    assert (np.diff(np.sort(m.row)) >= 0).all() and m.S.min() > 0.0
    lat_s, lon_s = np.meshgrid(src.lat, src.lon, indexing='ij')
    lat_d, lon_d = np.meshgrid(dst.lat, dst.lon, indexing='ij')
    count = np.bincount(m.row - 1, minlength=m.n_b).reshape(lat_d.shape)
    cap = np.abs(lat_d) > 85.0          # poleward of the last row of centres
    assert set(np.unique(count[~cap])) <= {1, 2, 4}
    assert np.all(count[cap] == len(src.lon))    # the whole row takes part
    P = _unit(np.radians(lat_s), np.radians(lon_s)).reshape(-1, 3)
    q = _unit(np.radians(lat_d), np.radians(lon_d)).reshape(-1, 3)
    R = A @ P
    R /= np.linalg.norm(R, axis=1)[:, None]
    assert np.abs(R - q)[~cap.reshape(-1)].max() < 1e-14
    # smooth fields to O(h^2); the edges of a quad are great circles, so not
    # even a field linear in latitude is reproduced exactly
    f = lambda la, lo: 2.0 + 0.1 * la + 3.0 * np.cos(np.radians(lo))
    got = (A @ f(lat_s, lon_s).reshape(-1)).reshape(lat_d.shape)
    assert np.abs(got - f(lat_d, lon_d))[~cap].max() < 0.04
    # at the pole itself: the mean of the last row
    from pyremap_amd import PointCollectionDescriptor
    polar = _dense(build_weights(
        src, PointCollectionDescriptor(np.array([90.0]), np.array([0.0]),
                                       'pole', units='degrees'), 'bilinear'))
    ring = f(lat_s, lon_s)[-1]
    assert abs((polar @ f(lat_s, lon_s).reshape(-1))[0] - ring.mean()) < 0.01


def _masked_apply(m, f):
    """`remap_numpy(ds, 0.01)` in numpy: NaNs masked, renormalised."""
    valid = ~np.isnan(f)
    num = np.bincount(m.row - 1, weights=m.S * np.where(valid, f, 0.0)[
        m.col - 1], minlength=m.n_b)
    den = np.bincount(m.row - 1, weights=m.S * valid[m.col - 1],
                      minlength=m.n_b)
    with np.errstate(all='ignore'):
        return np.where(den > 0.01, num / den, np.nan)


def test_rectangular_source_weights_reproduce_esmf_outputs():
    """
    Bilinear weights FROM the reference's 1-degree SST grid and from its
    100 km Antarctic stereographic grid, applied with numpy, against the
    outputs the reference stored for `test_latlon_file_to_latlon_array`,
    `test_latlon_to_mpas_cell` (+ `_expand`),
    `test_latlon_file_to_point_collection`, `test_latlon_to_stereographic` and
    `test_stereographic_array_to_latlon_array` (ESMF weights): equal to the
    rounding of the stored files -- float64 1e-11, float32 (written by NCO)
    6e-8 relative -- the point beyond the last latitude row (ESMF's pole
    cap) and the 13 720 unmapped cells of the last test included.
    """
    from pyremap_amd import (
        MpasCellMeshDescriptor,
        PointCollectionDescriptor,
        ProjectionGridDescriptor,
    )
    from pyremap_amd.io.netcdf import open_dataset
    from pyremap_amd.polar import get_antarctic_stereographic_projection
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                        'ref_fixtures')
    sst_file = os.path.join(here, 'SST_annual_1870-1900.nc')
    src = LatLonGridDescriptor.read(sst_file)
    assert src.regional is False
    sst = np.asarray(open_dataset(sst_file)['SST'].values[0],
                     dtype=np.float64).reshape(-1)

    def check(m, field, want, rtol):
        got = _masked_apply(m, field)
        want = np.asarray(want, dtype=np.float64).reshape(-1)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.abs(got - want)[ok].max() <= rtol * np.abs(want[ok]).max()

    ref = open_dataset(os.path.join(here,
                                    'ref_latlon_file_to_latlon_array.nc'))
    dst = LatLonGridDescriptor.create(np.linspace(-90.0, 90.0, 91),
                                      np.linspace(-180.0, 180.0, 181),
                                      units='degrees')
    check(build_weights(src, dst, 'bilinear'), sst, ref['SST'].values, 1e-11)
    ref = open_dataset(os.path.join(
        here, 'ref_latlon_file_to_point_collection.nc'))
    pts = PointCollectionDescriptor(ref['lat'].values, ref['lon'].values,
                                    'mpasCellCenters', units='degrees')
    m = build_weights(src, pts, 'bilinear')
    assert np.all(m.frac_b == 1.0)
    assert (np.abs(ref['lat'].values) > 89.5).sum() == 1    # the pole cap
    check(m, sst, ref['SST'].values, 1e-7)
    cells = MpasCellMeshDescriptor(os.path.join(here, 'mpasMesh.nc'),
                                   mesh_name='oQU240')
    m = build_weights(src, cells, 'bilinear')
    ref = open_dataset(os.path.join(os.path.dirname(here), 'hdf5',
                                    'nc4_ref_latlon_to_mpas_cell.nc'))
    check(m, sst, ref['SST'].values, 1e-11)
    ref = open_dataset(os.path.join(here,
                                    'ref_latlon_to_mpas_cell_expand.nc'))
    check(m, sst, ref['SST'].values, 1e-7)
    stereo = get_polar_descriptor(6000.0, 5000.0, 100.0, 100.0)
    ref = open_dataset(os.path.join(here, 'ref_latlon_to_stereographic.nc'))
    check(build_weights(src, stereo, 'bilinear'), sst, ref['SST'].values,
          1e-7)
    # the stereographic grid as the source
    x_max, y_max, res = 3000e3, 2500e3, 100e3
    grid = ProjectionGridDescriptor.create(
        get_antarctic_stereographic_projection(),
        np.linspace(-x_max, x_max, 61), np.linspace(-y_max, y_max, 51),
        '100km_Antarctic_stereo')
    ref = open_dataset(os.path.join(here, 'ref_stereographic_to_latlon.nc'))
    m = build_weights(grid, dst, 'bilinear')
    want = ref['complicated'].values[0, :, :, 0]
    assert np.isnan(want).sum() == 13720
    assert np.array_equal(m.frac_b == 0.0, np.isnan(want).reshape(-1))
    check(m, np.asarray(grid.coords['lat']['data']).reshape(-1), want, 1e-9)


def test_nearest_and_projection_grids():
    src = get_polar_descriptor(600.0, 400.0, 100.0, 100.0)
    dst = get_polar_descriptor(500.0, 300.0, 25.0, 25.0)
    for method in ('bilinear', 'neareststod', 'conserve'):
        m = build_weights(src, dst, method)
        A = _dense(m)
        assert np.allclose(A.sum(axis=1), m.frac_b, atol=1e-12), method
        assert np.allclose(m.frac_b, 1.0, atol=1e-12), method
        assert list(m.src_grid_dims) == [7, 5]
        assert list(m.dst_grid_dims) == [21, 13]
        xs, ys = np.meshgrid(src.x, src.y)
        xd, yd = np.meshgrid(dst.x, dst.y)
        if method == 'bilinear':
            # (on the sphere, not in the plane of the projection: a field
            # linear in x and y is reproduced to the grid's curvature)
            got = A @ (1.0 + 2e-5 * xs - 1e-5 * ys).reshape(-1)
            assert np.allclose(got, (1.0 + 2e-5 * xd - 1e-5 * yd).reshape(-1),
                               rtol=0, atol=3e-4)
        if method == 'neareststod':
            assert np.bincount(m.row - 1, minlength=m.n_b).max() == 1
    with pytest.raises(ValueError, match='conserve needs cells'):
        build_weights(src, get_lat_lon_descriptor(10.0, 10.0))
    with pytest.raises(ValueError, match='expected one of'):
        build_weights(src, dst, 'patch')


def test_write_weights_round_trip(tmp_path):
    src = get_lat_lon_descriptor(20.0, 20.0)
    dst = get_lat_lon_descriptor(15.0, 10.0)
    path = str(tmp_path / 'map.nc')
    m = write_weights(path, src, dst, 'conserve')
    back = mapfile.read_mapping(path)
    for name in ('row', 'col', 'S', 'frac_b', 'src_grid_dims',
                 'dst_grid_dims'):
        np.testing.assert_array_equal(getattr(back, name), getattr(m, name))
    assert (back.n_a, back.n_b) == (m.n_a, m.n_b)
    # Fortran order in the file: [nlon, nlat]
    assert list(back.src_grid_dims) == [len(src.lon), len(src.lat)]


def test_lat_lon_to_points():
    from pyremap_amd import MpasCellMeshDescriptor, PointCollectionDescriptor
    src = get_lat_lon_descriptor(5.0, 5.0)
    rng = np.random.default_rng(4)
    lat = np.degrees(np.arcsin(rng.uniform(-0.99, 0.99, 300)))
    lon = rng.uniform(-180.0, 180.0, 300)
    pts = PointCollectionDescriptor(lat, lon, 'pts', units='degrees')
    cells = MpasCellMeshDescriptor(mesh_name='m', lat=np.radians(lat),
                                   lon=np.radians(lon) % (2 * np.pi))
    f = lambda la, lo: 1.0 + 0.05 * la + np.cos(np.radians(lo))
    lat_s, lon_s = np.meshgrid(src.lat, src.lon, indexing='ij')
    for dst in (pts, cells):
        m = build_weights(src, dst, 'bilinear')
        assert list(m.dst_grid_dims) == [300] and m.n_b == 300
        A = _dense(m)
        assert np.allclose(A.sum(axis=1), 1.0)
        assert np.bincount(m.row - 1).max() <= 4
        got = A @ f(lat_s, lon_s).reshape(-1)
        inner = np.abs(lat) < 87.0
        assert np.abs(got - f(lat, lon))[inner].max() < 4e-3
        near = build_weights(src, dst, 'neareststod')
        assert near.n_s == 300 and np.all(near.S == 1.0)
        iy, ix = np.divmod(near.col - 1, len(src.lon))
        assert np.abs(src.lat[iy] - lat).max() <= 2.5 + 1e-9
        dlon = np.abs((src.lon[ix] - lon + 180.0) % 360.0 - 180.0)
        assert dlon.max() <= 2.5 + 1e-9
    # both descriptors describe the same points: same weights
    a = build_weights(src, pts, 'bilinear')
    b = build_weights(src, cells, 'bilinear')
    assert np.array_equal(a.col, b.col) and np.allclose(a.S, b.S, atol=1e-12)
    with pytest.raises(ValueError, match='conserve needs cells'):
        build_weights(src, pts, 'conserve')


def test_projection_grid_to_lat_lon_and_back():
    """Across grid kinds the destination cell centres are located in the
    source grid through the projection (the reference's
    test_stereographic_array_to_latlon_array / test_latlon_to_stereographic
    pairs): latitude, a smooth field of position, survives the trip."""
    stereo = get_polar_descriptor(6000.0, 5000.0, 100.0, 100.0)
    latlon = get_lat_lon_descriptor(2.0, 2.0)
    m = build_weights(stereo, latlon, 'bilinear')
    assert list(m.src_grid_dims) == [61, 51]
    assert list(m.dst_grid_dims) == [len(latlon.lon), len(latlon.lat)]
    A = _dense(m)
    mapped = m.frac_b > 0
    assert np.allclose(A.sum(axis=1)[mapped], 1.0)
    assert np.all(A.sum(axis=1)[~mapped] == 0.0)
    lat_src = stereo.coords['lat']['data'].reshape(-1)
    lat_dst = np.repeat(latlon.lat, len(latlon.lon))
    got = A @ lat_src
    assert 0.10 < mapped.mean() < 0.20          # the Antarctic cap only
    assert lat_dst[mapped].max() < -55.0
    assert np.abs(got - lat_dst)[mapped].max() < 0.12
    # and back: lat-lon -> stereographic (global source: every point mapped)
    back = build_weights(latlon, stereo, 'bilinear')
    assert np.all(back.frac_b == 1.0)
    B = _dense(back)
    lat_ll = np.repeat(latlon.lat, len(latlon.lon))
    # latitude comes back to the curvature of a 2-degree quad's edges;
    # poleward of the last row of centres (the pole itself sits in this
    # grid) the cap interpolates towards that row's mean
    err = np.abs(B @ lat_ll - lat_src)
    assert err[lat_src >= -89.0].max() < 0.01
    assert err.max() <= 1.0
    with pytest.raises(ValueError, match='conserve needs cells'):
        build_weights(stereo, latlon, 'conserve')


# ---------------------------------------------------------------------------
# an MPAS mesh as the source: ESMF's bilinear on the dual mesh
# ---------------------------------------------------------------------------

FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_clip_ears_partitions_convex_polygons():
    """n - 2 triangles per polygon, their areas add up to the polygon's, and
    the first ear cut off is the corner with the largest dot product of its
    two edge vectors."""
    from pyremap_amd.weights import clip_ears
    rng = np.random.default_rng(3)
    polys, counts, pts = [], [], []
    for n in (3, 5, 6, 7, 6, 4):
        ang = np.sort(rng.uniform(0.0, 2 * np.pi, n))
        rad = rng.uniform(0.8, 1.2, n)
        first = len(pts)
        for a, r in zip(ang, rad):
            pts.append((r * np.cos(a) * (2.0 if n == 7 else 1.0),
                        r * np.sin(a), 0.0))
        polys.append(list(range(first, first + n)) + [0] * (7 - n))
        counts.append(n)
    xyz = np.array(pts)
    # keep only convex ones (random radii can dent a polygon)
    def convex(ids):
        p = xyz[ids, :2]
        e = np.roll(p, -1, axis=0) - p
        cr = e[:, 0] * np.roll(e, -1, axis=0)[:, 1] - \
            e[:, 1] * np.roll(e, -1, axis=0)[:, 0]
        return (cr > 0).all()
    keep = [i for i, (p, n) in enumerate(zip(polys, counts))
            if convex(p[:n])]
    assert len(keep) >= 3
    polys = [polys[i] for i in keep]
    counts = [counts[i] for i in keep]
    tri = clip_ears(xyz, np.array(polys), np.array(counts))
    assert len(tri) == sum(n - 2 for n in counts)

    def area(ids):
        p = xyz[ids, :2]
        return 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) -
                                np.roll(p[:, 0], -1) * p[:, 1]))
    assert np.isclose(sum(area(t) for t in tri),
                      sum(area(p[:n]) for p, n in zip(polys, counts)))
    # the rule itself on a hexagon stretched along x: the first triangle cut
    # off sits at the corner with the largest edge-vector dot product
    hexa = np.array([(2, 0, 0), (1, 1, 0), (-1, 1, 0), (-2, 0, 0),
                     (-1, -1, 0), (1, -1, 0)], dtype=float)
    t = clip_ears(hexa, np.arange(6)[None, :], np.array([6]))
    dots = [(hexa[i - 1] - hexa[i]) @ (hexa[(i + 1) % 6] - hexa[i])
            for i in range(6)]
    assert set(t[0]) == {(int(np.argmax(dots)) - 1) % 6,
                         int(np.argmax(dots)),
                         (int(np.argmax(dots)) + 1) % 6}


@pytest.mark.parametrize('kind', ['cell', 'edge', 'vertex'])
def test_mpas_source_weights_reproduce_esmf_outputs(kind):
    """
    Bilinear weights FROM the reference's QU240 mesh (cells / edges /
    vertices) to its 1-degree grid, applied with numpy to the reference's
    input files, against the outputs the reference stored
    (`ref_mpas_{cell,edge,vertex}_to_latlon.nc`, made with ESMF weights): the
    same 22-26 thousand cells unmapped, every value equal to rounding.
    """
    from pyremap_amd import (
        MpasCellMeshDescriptor,
        MpasEdgeMeshDescriptor,
        MpasVertexMeshDescriptor,
    )
    from pyremap_amd.io.netcdf import open_dataset
    here = os.path.join(FIXTURES, 'ref_fixtures')
    cls, infile, names = {
        'cell': (MpasCellMeshDescriptor,
                 os.path.join(here, 'timeSeries.0002-01-01.nc'),
                 ('timeMonthly_avg_ssh', 'timeMonthly_avg_tThreshMLD')),
        'edge': (MpasEdgeMeshDescriptor,
                 os.path.join(here, 'mpasAreaEdge.nc'), ('areaEdge',)),
        'vertex': (MpasVertexMeshDescriptor,
                   os.path.join(FIXTURES, 'hdf5', 'nc4_mpasAreaVertex.nc'),
                   ('areaVertex',))}[kind]
    src = cls(os.path.join(here, 'mpasMesh.nc'), mesh_name='oQU240')
    dst = LatLonGridDescriptor.read(
        os.path.join(here, 'SST_annual_1870-1900.nc'))
    m = build_weights(src, dst, 'bilinear')
    assert m.n_a == src.dim_sizes[0] and m.n_b == 180 * 360
    assert list(m.src_grid_dims) == [m.n_a]
    assert list(m.dst_grid_dims) == [360, 180]
    assert set(np.unique(m.frac_b)) == {0.0, 1.0}
    # three corners per mapped point, weights in [0, 1] summing to one
    rows = np.bincount(m.row - 1, minlength=m.n_b)
    assert set(np.unique(rows)) == {0, 3}
    assert np.array_equal(rows > 0, m.frac_b > 0)
    assert m.S.min() >= 0.0 and m.S.max() <= 1.0
    sums = np.bincount(m.row - 1, weights=m.S, minlength=m.n_b)
    assert np.allclose(sums[rows > 0], 1.0, rtol=0, atol=1e-14)
    ds_in = open_dataset(infile)
    ds_ref = open_dataset(os.path.join(here, f'ref_mpas_{kind}_to_latlon.nc'))
    for name in names:
        f = np.asarray(ds_in[name].values, dtype=np.float64).reshape(-1)
        got = np.bincount(m.row - 1, weights=m.S * f[m.col - 1],
                          minlength=m.n_b)
        got[m.frac_b == 0.0] = np.nan
        want = ds_ref[name].values.reshape(-1)
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        ok = ~np.isnan(want)
        assert 20000 < (~ok).sum() < 27000
        assert np.isclose(got[ok], want[ok], rtol=1e-10, atol=1e-12).all()


def test_mpas_source_point_location_and_errors():
    """The located point is the central projection of the weighted corner
    sum (straight lines in 3-D); only bilinear, only with a mesh file."""
    from pyremap_amd import MpasCellMeshDescriptor, PointCollectionDescriptor
    from pyremap_amd.weights import _dual_triangles, _unit
    mesh = os.path.join(FIXTURES, 'ref_fixtures', 'mpasMesh.nc')
    src = MpasCellMeshDescriptor(mesh, mesh_name='oQU240')
    rng = np.random.default_rng(8)
    lat = np.arcsin(rng.uniform(-0.95, 0.95, 4000))
    lon = rng.uniform(0.0, 2 * np.pi, 4000)
    pts = PointCollectionDescriptor(lat, lon, 'pts', units='radians')
    m = build_weights(src, pts, 'bilinear')
    xyz, tri = _dual_triangles(src)
    assert len(tri) == 13317               # vertices with three ocean cells
    mapped = np.nonzero(m.frac_b > 0)[0]
    assert 0.5 < len(mapped) / 4000 < 0.8  # the ocean's share
    P = np.zeros((4000, 3))
    np.add.at(P, m.row - 1, m.S[:, None] * xyz[m.col - 1])
    q = _unit(lat, lon)
    P = P[mapped] / np.linalg.norm(P[mapped], axis=1)[:, None]
    assert np.abs(P - q[mapped]).max() < 1e-13
    with pytest.raises(ValueError, match='only bilinear'):
        build_weights(src, pts, 'conserve')
    bare = MpasCellMeshDescriptor(mesh_name='m', lat=lat, lon=lon)
    with pytest.raises(ValueError, match='mesh file'):
        build_weights(bare, pts, 'bilinear')


def test_bilinear_descending_shifted_and_regional_sources():
    """Source axes as files really hold them: latitudes north to south,
    longitudes from -180; and a regional source (no closure, no caps: points
    outside the hull of its cell centres stay unmapped)."""
    from pyremap_amd.weights import _unit
    dst = get_lat_lon_descriptor(2.0, 2.0)
    lat_d, lon_d = np.meshgrid(dst.lat, dst.lon, indexing='ij')
    f = lambda la, lo: np.sin(np.radians(la)) + \
        0.3 * np.cos(np.radians(la)) * np.cos(np.radians(lo))
    src = LatLonGridDescriptor.create(np.linspace(90.0, -90.0, 37),
                                      np.linspace(-180.0, 180.0, 73),
                                      units='degrees')
    assert src.regional is False and src.lat[0] > src.lat[-1]
    m = build_weights(src, dst, 'bilinear')
    A = _dense(m)
    assert np.all(m.frac_b == 1.0)
    lat_s, lon_s = np.meshgrid(src.lat, src.lon, indexing='ij')
    assert np.abs(A @ f(lat_s, lon_s).ravel() -
                  f(lat_d, lon_d).ravel()).max() < 2e-3
    count = np.bincount(m.row - 1, minlength=m.n_b)
    assert (count > 4).sum() == 2 * 180          # one capped row per pole
    P = _unit(np.radians(lat_s), np.radians(lon_s)).reshape(-1, 3)
    q = _unit(np.radians(lat_d), np.radians(lon_d)).reshape(-1, 3)
    R = A @ P
    R /= np.linalg.norm(R, axis=1)[:, None]
    assert np.abs(R - q)[count <= 4].max() < 1e-14
    # regional
    src = LatLonGridDescriptor.create(np.linspace(10.0, 50.0, 21),
                                      np.linspace(-30.0, 40.0, 36),
                                      units='degrees')
    assert src.regional is True
    m = build_weights(src, dst, 'bilinear')
    lon_w = (lon_d.ravel() + 180.0) % 360.0 - 180.0
    inside = (lat_d.ravel() >= src.lat.min()) & \
        (lat_d.ravel() <= src.lat.max()) & (lon_w >= src.lon.min()) & \
        (lon_w <= src.lon.max())
    assert np.array_equal(m.frac_b > 0, inside) and inside.sum() == 700
    assert np.bincount(m.row - 1, minlength=m.n_b).max() <= 4
    A = _dense(m)
    lat_s, lon_s = np.meshgrid(src.lat, src.lon, indexing='ij')
    got = A @ f(lat_s, lon_s).ravel()
    assert np.abs(got - f(lat_d.ravel(), lon_w))[inside].max() < 1e-3
