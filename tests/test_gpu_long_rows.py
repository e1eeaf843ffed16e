"""
Mappings whose few LONG rows hold a large share of the entries -- a global
lat-lon bilinear map as ESMF makes it: destination cells poleward of the last
source row take the whole adjacent source row (the pole cap,
`pyremap_amd.weights.bilinear_3d`), hundreds of entries per row among rows of
four -- are applied as two launches writing disjoint rows
(`RemapPlan._split_long_rows`): the mapping without those rows on its own
schedule, the long rows through the LDS-staged lanes-across-rows kernel on
column-major entries.  Every value against the oracle, bit for bit.
"""
import numpy as np
import pytest

from helpers import assert_bitwise

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from pyremap_amd import engine
    engine.load_library()
    return torch.device('cuda', 0)


def _capped_map():
    """2 deg -> 1 deg global bilinear, ESMF's way: 2 x 360 pole-cap rows
    holding the 180 cells of the adjacent source row."""
    from pyremap_amd import get_lat_lon_descriptor
    from pyremap_amd.weights import build_weights
    src = get_lat_lon_descriptor(2.0, 2.0)
    dst = get_lat_lon_descriptor(1.0, 1.0)
    m = build_weights(src, dst, 'bilinear')
    rows = np.bincount(m.row - 1, minlength=m.n_b)
    assert rows.max() == 180 and (rows > 8).sum() == 720
    return m, (180, 360)


def _ragged_long_map(seed=5):
    """Long rows WITHOUT any structure: 40 rows of 150-700 random columns
    with random (signed) weights among short conservative-like rows."""
    from pyremap_amd import synthetic
    m = synthetic.conservative_map(3000, (40, 50), 1, 6, seed=seed,
                                   locality='mesh')
    mm = m.numpy()
    rng = np.random.default_rng(seed)
    row, col, S = [mm['row']], [mm['col']], [mm['S']]
    for r in rng.choice(m.n_b, 40, replace=False):
        n = int(rng.integers(150, 700))
        row.append(np.full(n, r + 1))
        col.append(rng.choice(m.n_a, n, replace=False) + 1)
        S.append(rng.standard_normal(n))
    mm['row'] = np.concatenate(row).astype(np.int32)
    mm['col'] = np.concatenate(col).astype(np.int32)
    mm['S'] = np.concatenate(S)
    mm['frac_b'] = np.where(mm['frac_b'] > 0, mm['frac_b'], 0.5)
    return mm, m.n_a, m.n_b, (40, 50)


@pytest.mark.parametrize('which', ['pole caps', 'ragged'])
def test_long_rows_apart_bitwise(dev, which):
    from oracle import oracle
    from pyremap_amd import engine
    if which == 'pole caps':
        m, dims = _capped_map()
        row, col, S, frac_b, n_a, n_b = m.row, m.col, m.S, m.frac_b, \
            m.n_a, m.n_b
    else:
        mm, n_a, n_b, dims = _ragged_long_map()
        row, col, S, frac_b = mm['row'], mm['col'], mm['S'], mm['frac_b']
    plan = engine.RemapPlan.from_triplets(row, col, S, frac_b, n_a, n_b,
                                          device=dev)
    choice = plan.auto_schedule(dims)
    assert choice['long_rows'] == (720 if which == 'pole caps' else 40)
    assert choice['long_rows_layout'] == 'column-major'
    short, long = plan._split
    assert short.nnz + long.nnz == plan.nnz and short.max_row_nnz <= 96
    rowptr, c, v = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, c, v, (n_b, n_a))
    rng = np.random.default_rng(2)
    cases = [((n_a,), [0]), ((n_a, 3), [0]), ((n_a, 12), [0]),
             ((n_a, 64), [0]), ((n_a, 130), [0]), ((5, n_a), [1]),
             ((40, n_a), [1]), ((3, n_a, 10), [1]), ((2, n_a, 61), [1])]
    for shape, axes in cases:
        for dtype in (np.float64, np.float32):
            x = rng.standard_normal(shape).astype(dtype)
            holes = x.copy()
            dead = rng.random(n_a) < 0.2
            holes[(slice(None),) * axes[0] + (dead,)] = np.nan
            for field, thr in ((x, None), (holes, 0.3), (holes, None)):
                masked = thr is not None
                arg = np.ma.masked_array(field, np.isnan(field)) if masked \
                    else field
                want = oracle.remap_numpy_array(csr, frac_b, dims, arg, axes,
                                                thr)
                mask_out = None
                got = engine.remap_tensor(
                    plan, dims, torch.from_numpy(field).to(dev), axes,
                    engine.MODE_MASKED if masked else engine.MODE_FRACB,
                    threshold=thr or 0.0)
                assert_bitwise(got.cpu().numpy(), np.ma.filled(want, np.nan),
                               f'{which} {shape} {dtype.__name__} thr {thr}')
                del mask_out
    # the mask output, and a row range (no split there: the plain kernels)
    x = torch.from_numpy(rng.standard_normal((n_a, 24))).to(dev)
    y, mask = engine.remap_tensor(plan, dims, x, [0], engine.MODE_FRACB,
                                  want_mask=True)
    want, wmask = oracle.remap_flat(csr, frac_b, x.cpu().numpy(), False, 0.0)
    assert np.array_equal(mask.cpu().numpy().reshape(n_b, 24).astype(bool),
                          wmask)
    sub = plan.row_slice(100, 900)
    ys = engine.remap_tensor(sub, None, x, [0], engine.MODE_RAW)
    assert_bitwise(ys.cpu().numpy(),
                   oracle.csr_matvecs(csr, x.cpu().numpy())[100:900],
                   'row slice')


@pytest.mark.parametrize('tt', [1, 2, 4, 8, 16])
def test_wave_per_long_row_every_width(dev, tt, monkeypatch):
    """
    Kernel family 9 (one wave per long row and `tt` columns: entries read
    lanes-across-entries, summed in order from LDS) is what the long rows
    take for up to 16 fields; here it is forced for EVERY field count and
    layout, each column width: the oracle's bits.
    """
    from oracle import oracle
    from pyremap_amd import engine
    mm, n_a, n_b, dims = _ragged_long_map(seed=11)
    plan = engine.RemapPlan.from_triplets(mm['row'], mm['col'], mm['S'],
                                          mm['frac_b'], n_a, n_b, device=dev)
    plan.auto_schedule(dims)
    assert plan._split is not None
    monkeypatch.setattr(engine, 'LONG_WAVE_FIELDS', 1 << 30)
    monkeypatch.setattr(engine, '_LONG_WAVE_TT', tt)
    rowptr, c, v = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, c, v, (n_b, n_a))
    rng = np.random.default_rng(tt)
    for shape, axes in (((n_a,), [0]), ((n_a, 5), [0]), ((n_a, 37), [0]),
                        ((7, n_a), [1]), ((33, n_a), [1]),
                        ((3, n_a, 6), [1])):
        for dtype in (np.float64, np.float32):
            x = rng.standard_normal(shape).astype(dtype)
            holes = x.copy()
            holes[(slice(None),) * axes[0] + (rng.random(n_a) < 0.2,)] = \
                np.nan
            for field, thr in ((x, None), (holes, 0.3), (holes, None)):
                masked = thr is not None
                arg = np.ma.masked_array(field, np.isnan(field)) if masked \
                    else field
                want = oracle.remap_numpy_array(csr, mm['frac_b'], dims, arg,
                                                axes, thr)
                got, mask = engine.remap_tensor(
                    plan, dims, torch.from_numpy(field).to(dev), axes,
                    engine.MODE_MASKED if masked else engine.MODE_FRACB,
                    threshold=thr or 0.0, want_mask=True)
                assert_bitwise(got.cpu().numpy(), np.ma.filled(want, np.nan),
                               f'tt {tt} {shape} {dtype.__name__} thr {thr}')
                assert np.array_equal(mask.cpu().numpy().astype(bool),
                                      np.ma.getmaskarray(want))
    # two source axes with another dim between them (x_src_fold)
    lat, lon = 50, n_a // 50
    x = rng.standard_normal((lat, 3, lon))
    want = oracle.remap_numpy_array(csr, mm['frac_b'], dims, x, [0, 2], None)
    got = engine.remap_tensor(plan, dims, torch.from_numpy(x).to(dev),
                              [0, 2], engine.MODE_FRACB)
    assert_bitwise(got.cpu().numpy(), np.ma.filled(want, np.nan),
                   f'tt {tt} (lat, M, lon)')


@pytest.mark.parametrize('rows', [0, 1, 3, 8, 16])
@pytest.mark.parametrize('which', ['pole caps', 'ragged'])
def test_wave_per_long_row_windows_through_lds(dev, which, rows, monkeypatch):
    """
    Kernel family 11 (one wave per long row x 64 columns, the source cells of
    `rows` neighbouring long rows sliding through LDS in windows of 8 x rows
    cells) is what the long rows take beyond 16 fields; here for EVERY field
    count and layout, several patch heights (the ragged rows' unions span
    dozens of windows with a handful of entries each; the pole caps fill
    every window).  `rows` = 0: family 7 on column-major entries, the path
    of rounds 3 and 4 before.  The oracle's bits.
    """
    from oracle import oracle
    from pyremap_amd import engine
    monkeypatch.setattr(engine, 'LONG_WAVE_ROWS', rows)
    if which == 'pole caps':
        m, dims = _capped_map()
        row, col, S, frac_b, n_a, n_b = m.row, m.col, m.S, m.frac_b, \
            m.n_a, m.n_b
    else:
        mm, n_a, n_b, dims = _ragged_long_map(seed=23)
        row, col, S, frac_b = mm['row'], mm['col'], mm['S'], mm['frac_b']
    plan = engine.RemapPlan.from_triplets(row, col, S, frac_b, n_a, n_b,
                                          device=dev)
    plan.auto_schedule(dims)
    long = plan._split[1]
    if rows:
        assert 1 <= long._wave['rows'] <= rows   # (what the LDS holds)
        monkeypatch.setattr(engine, 'LONG_WAVE_FIELDS', 0)
        monkeypatch.setattr(engine, 'LONG_WAVE_MAX', 1 << 30)
    else:
        assert long._wave is None
    rowptr, c, v = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, c, v, (n_b, n_a))
    rng = np.random.default_rng(rows)
    for shape, axes in (((n_a,), [0]), ((n_a, 17), [0]), ((n_a, 64), [0]),
                        ((n_a, 200), [0]), ((70, n_a), [1]),
                        ((3, n_a, 61), [1]), ((2, n_a, 128), [1])):
        for dtype in (np.float64, np.float32):
            x = rng.standard_normal(shape).astype(dtype)
            holes = x.copy()
            holes[(slice(None),) * axes[0] + (rng.random(n_a) < 0.2,)] = \
                np.nan
            for field, thr in ((x, None), (holes, 0.3), (holes, None)):
                masked = thr is not None
                arg = np.ma.masked_array(field, np.isnan(field)) if masked \
                    else field
                want = oracle.remap_numpy_array(csr, frac_b, dims, arg, axes,
                                                thr)
                got, mask = engine.remap_tensor(
                    plan, dims, torch.from_numpy(field).to(dev), axes,
                    engine.MODE_MASKED if masked else engine.MODE_FRACB,
                    threshold=thr or 0.0, want_mask=True)
                assert_bitwise(got.cpu().numpy(), np.ma.filled(want, np.nan),
                               f'{rows} rows {shape} {dtype.__name__} {thr}')
                assert np.array_equal(mask.cpu().numpy().astype(bool),
                                      np.ma.getmaskarray(want))
    # two source axes with another dim between them (x_src_fold)
    lat = 50 if which == 'ragged' else dims[0] // 2
    x = rng.standard_normal((lat, 40, n_a // lat))
    want = oracle.remap_numpy_array(csr, frac_b, dims, x, [0, 2], None)
    got = engine.remap_tensor(plan, dims, torch.from_numpy(x).to(dev),
                              [0, 2], engine.MODE_FRACB)
    assert_bitwise(got.cpu().numpy(), np.ma.filled(want, np.nan),
                   f'{rows} rows (lat, M, lon)')
    # REMAP_FLAG_FMA: fused multiply-adds, within rounding of the oracle
    x = rng.standard_normal((n_a, 96))
    want = oracle.remap_numpy_array(csr, frac_b, dims, x, [0], None)
    got = engine.remap_tensor(plan, dims, torch.from_numpy(x).to(dev), [0],
                              engine.MODE_FRACB, flags=engine.FLAG_FMA)
    np.testing.assert_allclose(got.cpu().numpy(), np.ma.filled(want, np.nan),
                               rtol=1e-12, atol=1e-13)


def test_bench_pole_cap_map_at_full_size(dev, monkeypatch):
    """`config1_esmf` (the bench's 1 deg -> 0.5 deg map with pole caps: 1 440
    rows of 360 entries) at its full size, the field counts of the bench
    rows and their neighbours: the default route (families 9 / 11 / 7 for
    the long rows by field count) against the oracle, bit for bit, and --
    the same bits -- with family 11 switched off."""
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.make_config('config1_esmf', device=dev)
    plans = []
    for rows in (engine.LONG_WAVE_ROWS, 0):
        monkeypatch.setattr(engine, 'LONG_WAVE_ROWS', rows)
        plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                              m.n_a, m.n_b, index_base=1,
                                              device=dev)
        choice = plan.auto_schedule(m.dst_dims)
        assert choice['long_rows'] == 1440
        plans.append(plan)
    assert plans[0]._split[1]._wave['rows'] == 6
    assert plans[1]._split[1]._wave is None
    rowptr, c, v = plans[0].to_host_csr()
    csr = oracle.OracleCSR(rowptr, c, v, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    rng = np.random.default_rng(0)
    for shape, axes in (((m.n_a,), [0]), ((m.n_a, 17), [0]),
                        ((m.n_a, 64), [0]), ((m.n_a, 128), [0]),
                        ((m.n_a, 129), [0]), ((12, m.n_a), [1]),
                        ((2, m.n_a, 60), [1])):
        x = rng.standard_normal(shape)
        x[(slice(None),) * axes[0] + (rng.random(m.n_a) < 0.1,)] = np.nan
        for thr in (None, 0.2):
            arg = x if thr is None else np.ma.masked_array(x, np.isnan(x))
            want = np.ma.filled(oracle.remap_numpy_array(
                csr, frac_b, m.dst_dims, arg, axes, thr), np.nan)
            for plan in plans:
                got = engine.remap_tensor(
                    plan, m.dst_dims, torch.from_numpy(x).to(dev), axes,
                    engine.MODE_FRACB if thr is None else engine.MODE_MASKED,
                    threshold=thr or 0.0)
                assert_bitwise(got.cpu().numpy(), want, f'{shape} thr {thr}')


def test_pole_capped_map_through_the_remapper(dev, tmp_path):
    """build_map (ESMF's bilinear, pole caps and all) -> remap_numpy /
    ncremap: the Dataset path on a split plan, against the oracle; the
    auto-mode branch (NaN scan + gated launches) takes both launches."""
    from oracle import oracle
    from pyremap_amd import (
        DataArray,
        Dataset,
        Remapper,
        get_lat_lon_descriptor,
    )
    from pyremap_amd.io.netcdf import open_dataset, write_netcdf
    src = get_lat_lon_descriptor(2.0, 2.0)
    dst = get_lat_lon_descriptor(1.0, 1.0)
    r = Remapper(map_filename=str(tmp_path / 'map.nc'), method='bilinear',
                 map_tool='analytic', src_descriptor=src,
                 dst_descriptor=dst)
    r.build_map()
    plan = r.load_mapping()
    assert plan._split is not None
    rowptr, c, v = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, c, v, (plan.n_b, plan.n_a))
    frac_b = plan.frac_b.cpu().numpy()
    rng = np.random.default_rng(4)
    sst = rng.standard_normal((90, 180))
    land = sst.copy()
    land[20:40, 60:100] = np.nan
    monthly = rng.standard_normal((12, 90, 180)).astype(np.float32)
    ds = Dataset()
    ds['sst'] = DataArray(sst, dims=('lat', 'lon'))
    ds['land'] = DataArray(land, dims=('lat', 'lon'))
    ds['monthly'] = DataArray(monthly, dims=('time', 'lat', 'lon'))
    out = r.remap_numpy(ds, 0.05)
    for name, field, axes in (('sst', sst, [0, 1]), ('land', land, [0, 1]),
                              ('monthly', monthly, [1, 2])):
        nan = np.isnan(field).any()
        arg = np.ma.masked_array(field, np.isnan(field)) if nan else field
        want = oracle.remap_numpy_array(csr, frac_b, (180, 360), arg, axes,
                                        0.05)
        assert_bitwise(out[name].values, np.ma.filled(want, np.nan), name)
    # a smooth field survives, the poles included (the cap interpolates
    # towards the mean of the last row)
    lat = np.deg2rad(np.asarray(src.lat))[:, None]
    lon = np.deg2rad(np.asarray(src.lon))[None, :]
    smooth = np.sin(lat) + 0.3 * np.cos(lat) * np.cos(lon)
    got = r.remap_numpy(DataArray(smooth, dims=('lat', 'lon'))).values
    lat_d = np.deg2rad(np.asarray(dst.lat))[:, None]
    lon_d = np.deg2rad(np.asarray(dst.lon))[None, :]
    assert np.abs(got - (np.sin(lat_d) + 0.3 * np.cos(lat_d) *
                         np.cos(lon_d))).max() < 0.01
    write_netcdf(ds, str(tmp_path / 'in.nc'))
    r.ncremap(str(tmp_path / 'in.nc'), str(tmp_path / 'out.nc'),
              renormalize=0.05)
    back = open_dataset(str(tmp_path / 'out.nc'))
    for name in ('sst', 'land', 'monthly'):
        assert_bitwise(back[name].values, out[name].values, name)


def test_split_plan_on_devices_shards_and_graphs(dev):
    """Shards of a capped map (row ranges: unsplit), the single-process
    multi-device front end, and a captured series -- same bits."""
    from pyremap_amd import engine
    from pyremap_amd.parallel import MultiDeviceRemap
    m, dims = _capped_map()
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    plan.auto_schedule(dims)
    x = torch.randn((6, m.n_a, 5), device=dev, dtype=torch.float64)
    want = engine.remap_tensor(plan, dims, x, [1], engine.MODE_FRACB,
                               tune=[1])            # the plain kernel
    got = engine.remap_tensor(plan, dims, x, [1], engine.MODE_FRACB)
    assert torch.equal(got, want)
    multi = MultiDeviceRemap(plan, [dev, dev, dev], grid_dims=dims)
    # the first and the last shard hold a pole cap each: split like the whole
    assert multi.shards[0].plan._split is not None
    assert multi.shards[1].plan._split is None
    assert multi.shards[2].plan._split is not None
    assert torch.equal(engine.remap_tensor(multi, dims, x, [1],
                                           engine.MODE_FRACB), want)
    y = torch.empty_like(want)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        engine.remap_tensor(plan, dims, x, [1], engine.MODE_FRACB, out=y)
    x.copy_(torch.randn_like(x))
    y.fill_(-1.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, engine.remap_tensor(plan, dims, x, [1],
                                              engine.MODE_FRACB, tune=[1]))


@pytest.mark.parametrize('which', ['pole caps', 'ragged'])
def test_plan_handle_applies_long_rows_apart(dev, which):
    """The C plan handle (`remap_plan_create` / `_apply`) makes the same
    split inside the library: host triplets in, two launches per apply, the
    oracle's bits out -- (n_a, K), (Time, n_a) before and after
    `remap_plan_prepare_short_runs`, float32, masked."""
    import ctypes

    from oracle import oracle
    from pyremap_amd import engine
    lib = engine.load_library()
    if which == 'pole caps':
        m, dims = _capped_map()
        row, col, S, frac_b, n_a, n_b = m.row, m.col, m.S, m.frac_b, \
            m.n_a, m.n_b
    else:
        mm, n_a, n_b, dims = _ragged_long_map(seed=7)
        row, col, S, frac_b = mm['row'], mm['col'], mm['S'], mm['frac_b']
    csr = oracle.coo_to_csr(np.asarray(row) - 1, np.asarray(col) - 1,
                            np.asarray(S), n_b, n_a)

    def ptr(a):
        return a.ctypes.data_as(ctypes.c_void_p)
    row32 = np.ascontiguousarray(row, dtype=np.int32)
    col32 = np.ascontiguousarray(col, dtype=np.int32)
    S64 = np.ascontiguousarray(S, dtype=np.float64)
    fb = np.ascontiguousarray(frac_b, dtype=np.float64)
    cdims = (ctypes.c_int64 * 2)(*dims)
    handle = ctypes.c_void_p()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.remap_plan_create(n_b, n_a, S64.size, ptr(row32), ptr(col32),
                               ptr(S64), 1, ptr(fb), 1, cdims, 2, stream,
                               ctypes.byref(handle))
    assert rc == 0, lib.remap_last_error()
    try:
        info = engine._PlanInfo()
        assert lib.remap_plan_query(handle, ctypes.byref(info)) == 0
        assert info.nnz == len(csr.data)
        assert info.max_row_nnz == int(np.diff(csr.indptr).max()) > 96
        assert info.family in (5, 10)     # the REST got a real schedule
        rng = np.random.default_rng(6)
        for prepared in (False, True):
            if prepared:
                assert lib.remap_plan_prepare_short_runs(handle, stream) == 0
            # (1, 12 fields: family 9; 30 ... 128: family 11; 200: family 7)
            for shape, axis in (((n_a, 1), 0), ((n_a, 12), 0),
                                ((n_a, 200), 0), ((30, n_a), 1),
                                ((4, n_a, 9), 1), ((n_a, 64), 0),
                                ((2, n_a, 64), 1)):
                for dtype in (np.float64, np.float32):
                    x = rng.standard_normal(shape).astype(dtype)
                    x[(slice(None),) * axis + (rng.random(n_a) < 0.2,)] = \
                        np.nan
                    for masked in (False, True):
                        arg = np.ma.masked_array(x, np.isnan(x)) if masked \
                            else x
                        want = np.ma.filled(oracle.remap_numpy_array(
                            csr, fb, dims, arg, [axis],
                            0.2 if masked else None), np.nan)
                        X = torch.from_numpy(x).to(dev)
                        lead = shape[:axis]
                        tail = shape[axis + 1:]
                        Y = torch.full(lead + (n_b,) + tail, 3.0,
                                       dtype=torch.float64, device=dev)
                        f = engine._Field()
                        f.X, f.Y = X.data_ptr(), Y.data_ptr()
                        f.x_dtype = engine.DTYPE_F64 \
                            if dtype == np.float64 else engine.DTYPE_F32
                        f.mode = engine.MODE_MASKED if masked \
                            else engine.MODE_FRACB
                        f.threshold = 0.2
                        inner = int(np.prod(tail)) if tail else 1
                        f.n_batch = int(np.prod(lead)) if lead else 1
                        f.k_inner = inner
                        f.x_row_stride = f.y_row_stride = inner
                        f.x_batch_stride = n_a * inner
                        f.y_batch_stride = n_b * inner
                        s = ctypes.c_void_p(
                            torch.cuda.current_stream().cuda_stream)
                        rc = lib.remap_plan_apply(handle, ctypes.byref(f), s)
                        assert rc == 0, lib.remap_last_error()
                        assert_bitwise(
                            Y.cpu().numpy().reshape(want.shape), want,
                            f'{which} {shape} {dtype.__name__} masked '
                            f'{masked} prepared {prepared}')
    finally:
        lib.remap_plan_destroy(handle)


def test_unsorted_long_rows_keep_the_lanes_across_rows_kernel(dev):
    """Family 11 walks a row's entries window by window and needs them sorted
    by source cell.  A CSR handed over with UNSORTED rows (a caller's own
    arrays: `RemapPlan(...)` takes them as they are; scipy sums in the order
    it is given, so do the kernels) gets no family-11 plan: the long rows
    stay on family 7, and the results are the oracle's on that very order."""
    from oracle import oracle
    from pyremap_amd import engine
    mm, n_a, n_b, dims = _ragged_long_map(seed=31)
    sorted_plan = engine.RemapPlan.from_triplets(
        mm['row'], mm['col'], mm['S'], mm['frac_b'], n_a, n_b, device=dev)
    rowptr, col, val = (t.copy() for t in sorted_plan.to_host_csr())
    rng = np.random.default_rng(1)
    for r in range(n_b):       # shuffle inside every long row
        s, e = int(rowptr[r]), int(rowptr[r + 1])
        if e - s > 96:
            perm = rng.permutation(e - s)
            col[s:e] = col[s:e][perm]
            val[s:e] = val[s:e][perm]
    pad = engine.CSR_PAD
    plan = engine.RemapPlan(
        n_a, n_b, torch.from_numpy(rowptr).to(dev),
        torch.from_numpy(np.concatenate([col, np.zeros(pad, col.dtype)]))
        .to(dev),
        torch.from_numpy(np.concatenate([val, np.zeros(pad)])).to(dev),
        torch.from_numpy(mm['frac_b']).to(dev))
    plan.auto_schedule(dims)
    assert plan._split is not None and plan._split[1]._wave is None
    assert sorted_plan.auto_schedule(dims) and \
        sorted_plan._split[1]._wave is not None
    csr = oracle.OracleCSR(rowptr, col, val, (n_b, n_a))
    x = rng.standard_normal((n_a, 64))
    want = np.ma.filled(oracle.remap_numpy_array(csr, mm['frac_b'], dims, x,
                                                 [0], None), np.nan)
    got = engine.remap_tensor(plan, dims, torch.from_numpy(x).to(dev), [0],
                              engine.MODE_FRACB)
    assert_bitwise(got.cpu().numpy(), want, 'unsorted long rows')
