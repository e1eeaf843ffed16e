"""
(Time, nCells) and the other layouts whose contiguous run behind the source
axes is short -- the reference's most common input
(tests/test_interpolate.py:57-59, flattened by a transpose copy at
remap_numpy.py:254-256) -- addressed IN PLACE for any number of time slices
(VERDICT round 2, item 7): the lanes-across-rows kernels `spmm_rowcell`
(family 4) and `spmm_patchcell` (family 7, LDS-staged, on a patch plan the
plan builds on first use), bitwise against the oracle.
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


def _problem(dev, grid, locality='mesh', n_a=6000, seed=9):
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    dims = grid if len(grid) == 2 else (1, grid[0])
    m = synthetic.conservative_map(n_a, dims, 1, 7, seed=seed, device=dev,
                                   locality=locality)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    choice = plan.auto_schedule(grid)
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    return m, plan, csr, m.frac_b.cpu().numpy(), choice


SHAPES = [((40, 6000), [1]), ((120, 6000), [1]), ((7, 6000, 3), [1]),
          ((33, 6000, 1), [1]), ((5, 8, 6000), [2]), ((12, 6000), [1]),
          ((2, 6000), [1]), ((3, 6000, 7), [1]), ((65, 6000), [1]),
          ((2, 2, 6000, 2, 2), [2])]


@pytest.mark.parametrize('grid', [(50, 80), (4000,)], ids=['2d', '1d'])
@pytest.mark.parametrize('mode', ['fracb', 'masked', 'raw'])
def test_short_run_layouts_bitwise(dev, grid, mode):
    from oracle import oracle
    from pyremap_amd import engine
    m, plan, csr, frac_b, _ = _problem(dev, grid)
    emode = {'fracb': engine.MODE_FRACB, 'masked': engine.MODE_MASKED,
             'raw': engine.MODE_RAW}[mode]
    rng = np.random.default_rng(5)
    for shape, axes in SHAPES:
        for dtype in (np.float64, np.float32):
            x = rng.standard_normal(shape).astype(dtype)
            if mode == 'masked':
                dead = rng.random(m.n_a) < 0.2
                x[(slice(None),) * axes[0] + (dead,)] = np.nan
                x[rng.random(shape) < 0.05] = np.nan
            X = np.moveaxis(x, axes[0], 0).reshape(m.n_a, -1)
            # (raw = the bare product: frac_b = 1 divides by nothing)
            ref, ref_mask = oracle.remap_flat(
                csr, np.ones_like(frac_b) if mode == 'raw' else frac_b, X,
                mode == 'masked', 0.05)
            ref = ref.copy()
            ref[ref_mask] = np.nan
            lead = list(shape[:axes[0]])
            tail = list(shape[axes[0] + 1:])
            want = np.moveaxis(ref.reshape([m.n_b] + lead + tail), 0,
                               axes[0])
            want = want.reshape(lead + list(grid) + tail)
            xd = torch.from_numpy(x).to(dev)
            for tune in (None, [4], [4, 4, 4], [4, 16, 1], [7, 4], [7, 8],
                         [7, 16], [2], [1]):
                K = x.size // m.n_a
                if tune == [2] and K > 32:
                    continue
                if tune and tune[0] == 7:
                    assert plan.cell_patches() is not None
                    args_tune = tune
                    # explicit family 7: through the plan's cell patch plan
                    y, mask = _apply_cell(plan, grid, xd, axes, emode,
                                          args_tune)
                else:
                    y, mask = engine.remap_tensor(
                        plan, grid, xd, axes, emode, threshold=0.05,
                        want_mask=True, tune=tune)
                assert tuple(y.shape) == want.shape
                assert_bitwise(y.cpu().numpy(), want,
                               f'{shape} {dtype.__name__} {mode} {tune}')
                want_mask = np.moveaxis(
                    ref_mask.reshape([m.n_b] + lead + tail), 0,
                    axes[0]).reshape(want.shape)
                assert np.array_equal(mask.cpu().numpy().astype(bool),
                                      want_mask), (shape, mode, tune)


def _apply_cell(plan, grid, xd, axes, emode, tune):
    """`remap_tensor`'s own route for short runs, with an explicit TT."""
    from pyremap_amd import engine
    lead = axes[0]
    n_batch = int(np.prod(xd.shape[:lead], dtype=np.int64))
    k_inner = int(np.prod(xd.shape[lead + 1:], dtype=np.int64))
    out_shape = list(xd.shape[:lead]) + list(grid) + list(xd.shape[lead + 1:])
    y = torch.empty(out_shape, dtype=torch.float64, device=xd.device)
    mask = torch.empty(out_shape, dtype=torch.uint8, device=xd.device)
    args = plan._prefilled(True, True)
    real = plan._prefilled
    plan._prefilled = lambda whole, cell=False: real(whole, True)
    try:
        engine.apply_strided(
            plan, xd, y, n_batch=n_batch, k_inner=k_inner,
            x_row_stride=k_inner, x_batch_stride=plan.n_a * k_inner,
            y_row_stride=k_inner, y_batch_stride=plan.n_b * k_inner,
            mode=emode, threshold=0.05, mask_out=mask, tune=tune)
    finally:
        plan._prefilled = real
    del args
    return y, mask


def test_time_ncells_takes_one_launch_no_copy(dev, monkeypatch):
    """(Time = 120, nCells): ONE launch on the caller's tensor, no permute
    copy, through the LDS-staged kernel; partial row ranges and plans
    without a grid take `spmm_rowcell`."""
    from oracle import oracle
    from pyremap_amd import engine
    m, plan, csr, frac_b, choice = _problem(dev, (50, 80))
    assert choice['family'] == 'rowgroup'
    calls = []
    real = engine.apply_strided

    def spy(plan_, X, Y, **kw):
        calls.append((X.data_ptr(), kw['n_batch'], kw['k_inner'],
                      kw['x_row_stride'], kw['x_batch_stride']))
        return real(plan_, X, Y, **kw)
    monkeypatch.setattr(engine, 'apply_strided', spy)
    x = torch.randn((120, m.n_a), dtype=torch.float64, device=dev)
    y = engine.remap_tensor(plan, m.dst_dims, x, [1], engine.MODE_FRACB)
    assert calls == [(x.data_ptr(), 120, 1, 1, m.n_a)]
    assert plan._cell and plan._cell['rows'] <= 1024
    ref, ref_mask = oracle.remap_flat(csr, frac_b, x.cpu().numpy().T.copy(),
                                      False, 0.0)
    ref[ref_mask] = np.nan
    assert_bitwise(y.cpu().numpy().reshape(120, m.n_b), ref.T, 'T120')
    # the masked / unmasked decision on the device, gated launches
    x[:, ::7] = float('nan')
    y = engine.remap_tensor_auto_mode(plan, m.dst_dims, x, [1], 0.05)
    ref, ref_mask = oracle.remap_flat(csr, frac_b, x.cpu().numpy().T.copy(),
                                      True, 0.05)
    ref[ref_mask] = np.nan
    assert_bitwise(y.cpu().numpy().reshape(120, m.n_b), ref.T, 'T120 auto')
    # a row shard of the same plan: partial ranges run without patches
    monkeypatch.setattr(engine, 'apply_strided', real)
    sh = plan.row_slice(1000, 3000)
    ys = engine.remap_tensor(sh, None, x, [1], engine.MODE_MASKED,
                             threshold=0.05)
    assert_bitwise(ys.cpu().numpy(), ref.T[:, 1000:3000], 'shard')


@pytest.mark.parametrize('locality', ['raster', 'scatter'])
def test_time_ncells_other_numberings(dev, locality):
    from oracle import oracle
    from pyremap_amd import engine
    m, plan, csr, frac_b, _ = _problem(dev, (50, 80), locality=locality,
                                       seed=2)
    x = torch.randn((70, m.n_a), dtype=torch.float64, device=dev)
    y = engine.remap_tensor(plan, m.dst_dims, x, [1], engine.MODE_FRACB)
    ref, ref_mask = oracle.remap_flat(csr, frac_b, x.cpu().numpy().T.copy(),
                                      False, 0.0)
    ref[ref_mask] = np.nan
    assert_bitwise(y.cpu().numpy().reshape(70, m.n_b), ref.T, locality)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_time_level_lat_lon_on_a_bilinear_map(dev, dtype):
    """
    Lat-lon model output, `(time, lev, lat, lon)` remapped over its last two
    axes (the reference's real lat-lon input is f32 `(time, lat, lon)`,
    tests/test_interpolate.py:492-516): time x lev batches of ONE field each
    -- the lanes-across-rows kernel on its own patch plan, beside the LDS
    patch schedule the bilinear map itself gets (family 5).
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.bilinear_map((30, 40), (70, 90), device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    assert plan.auto_schedule(m.dst_dims)['family'] == 'patch'
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    rng = np.random.default_rng(31)
    for shape, axes in (((6, 5, 30, 40), [2, 3]), ((40, 30, 40), [1, 2]),
                        ((2, 30, 40), [1, 2])):
        x = rng.standard_normal(shape).astype(dtype)
        x[..., 3:9, 10:20] = np.nan            # land
        for thr in (None, 0.3):
            arg = x if thr is None else np.ma.masked_array(x, np.isnan(x))
            ref = np.ma.filled(oracle.remap_numpy_array(
                csr, frac_b, m.dst_dims, arg, axes, thr), np.nan)
            y = engine.remap_tensor(
                plan, m.dst_dims, torch.from_numpy(x).to(dev), axes,
                engine.MODE_FRACB if thr is None else engine.MODE_MASKED,
                threshold=thr or 0.0)
            assert_bitwise(y.cpu().numpy(), ref, f'{shape} {thr}')
    assert plan._cell and plan.patches is not None   # both plans live
    # ... and the (n, K) layout still takes the map's own LDS patches
    xk = rng.standard_normal((m.n_a, 96)).astype(dtype)
    ref = np.ma.filled(oracle.remap_numpy_array(
        csr, frac_b, m.dst_dims, xk, [0], None), np.nan)
    y = engine.remap_tensor(plan, m.dst_dims, torch.from_numpy(xk).to(dev),
                            [0], engine.MODE_FRACB)
    assert_bitwise(y.cpu().numpy(), ref, 'nk after the cell plan')


def test_non_adjacent_source_axes_in_place(dev, monkeypatch):
    """
    Two source axes with other dims between them -- `(lat, M, lon[, T])`,
    the last layout the reference's transpose copy (remap_numpy.py:254-256)
    was still needed for: addressed in place through two source strides
    (`remap_apply_args.x_src_fold`) by the lanes-across-rows kernels, one
    launch per index of the dims in front, no permute copy; bitwise.
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.bilinear_map((30, 40), (70, 90), device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    plan.auto_schedule(m.dst_dims)
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    calls = []
    real = engine.apply_strided

    def spy(plan_, X, Y, **kw):
        calls.append((kw['n_batch'], kw['k_inner'], kw.get('x_src_fold', 0)))
        return real(plan_, X, Y, **kw)
    monkeypatch.setattr(engine, 'apply_strided', spy)
    monkeypatch.setattr(torch.Tensor, 'permute',
                        lambda *a, **k: (_ for _ in ()).throw(
                            AssertionError('permute copy taken')))
    rng = np.random.default_rng(77)
    for shape, axes in (((30, 7, 40), [0, 2]), ((3, 30, 5, 40, 2), [1, 3]),
                        ((30, 2, 3, 40), [0, 3]), ((2, 30, 50, 40, 9), [1, 3]),
                        ((30, 1, 40), [0, 2])):
        for dtype in (np.float64, np.float32):
            x = rng.standard_normal(shape).astype(dtype)
            x[rng.random(shape) < 0.1] = np.nan
            for thr in (None, 0.2):
                arg = x if thr is None else np.ma.masked_array(x, np.isnan(x))
                ref = oracle.remap_numpy_array(csr, frac_b, m.dst_dims, arg,
                                               axes, thr)
                want_mask = np.ma.getmaskarray(ref)
                ref = np.ma.filled(ref, np.nan)
                calls.clear()
                y, mask = engine.remap_tensor(
                    plan, m.dst_dims, torch.from_numpy(x).to(dev), axes,
                    engine.MODE_FRACB if thr is None else engine.MODE_MASKED,
                    threshold=thr or 0.0, want_mask=True)
                assert tuple(y.shape) == ref.shape, (shape, axes)
                assert_bitwise(y.cpu().numpy(), ref,
                               f'{shape} {axes} {dtype.__name__} {thr}')
                assert np.array_equal(mask.cpu().numpy().astype(bool),
                                      want_mask)
                lead = int(np.prod(shape[:axes[0]], dtype=np.int64))
                assert len(calls) == lead and all(c[2] == 40 for c in calls)
                # numpy in -> numpy out (the dims BETWEEN the source axes
                # keep their place behind the destination dims)
                from pyremap_amd import host_path
                got, gmask = host_path.remap_host_array(
                    plan, m.dst_dims, x, axes,
                    mode='fracb' if thr is None else 'masked',
                    threshold=thr, want_mask=True).result()
                assert_bitwise(got, ref, f'host {shape} {axes}')
                assert np.array_equal(gmask, want_mask)


def test_empty_shapes_through_every_layout(dev):
    """Empty inputs (the reference's edge cases: no time slices, no levels,
    nothing between two source axes) give empty outputs of the right shape
    on every layout path, without a launch and without an error."""
    from pyremap_amd import DataArray, Dataset, Remapper, engine, synthetic
    m = synthetic.conservative_map(3000, (30, 40), 1, 6, seed=3, device=dev,
                                   locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    plan.auto_schedule(m.dst_dims)
    for shape, axes, want in (((0, 3000), [1], (0, 30, 40)),
                              ((2, 3000, 0), [1], (2, 30, 40, 0)),
                              ((0, 3000, 5), [1], (0, 30, 40, 5)),
                              ((3000, 0), [0], (30, 40, 0)),
                              ((5, 0, 3000), [2], (5, 0, 30, 40))):
        x = torch.randn(shape, device=dev, dtype=torch.float64)
        for mode in (engine.MODE_FRACB, engine.MODE_MASKED):
            y, mask = engine.remap_tensor(plan, m.dst_dims, x, axes, mode,
                                          threshold=0.1, want_mask=True)
            assert tuple(y.shape) == tuple(mask.shape) == want
    mb = synthetic.bilinear_map((10, 12), (20, 24), device=dev)
    pb = engine.RemapPlan.from_triplets(mb.row, mb.col, mb.S, mb.frac_b,
                                        mb.n_a, mb.n_b, device=dev)
    pb.auto_schedule(mb.dst_dims)
    for shape, axes, want in (((10, 0, 12), [0, 2], (20, 24, 0)),
                              ((0, 10, 3, 12), [1, 3], (0, 20, 24, 3)),
                              ((2, 10, 3, 12, 0), [1, 3], (2, 20, 24, 3, 0))):
        x = torch.randn(shape, device=dev, dtype=torch.float64)
        y = engine.remap_tensor(pb, mb.dst_dims, x, axes, engine.MODE_FRACB)
        assert tuple(y.shape) == want
    torch.cuda.synchronize()

    class Desc:
        pass
    s, d = Desc(), Desc()
    s.dims, s.dim_sizes = ['nCells'], [m.n_a]
    d.dims, d.dim_sizes = ['lat', 'lon'], list(m.dst_dims)
    d.coords, d.mesh_name = {}, 'g'
    mm = m.numpy()
    r = Remapper.from_triplets(mm['row'], mm['col'], mm['S'], mm['frac_b'],
                               s, d, device=dev)
    out = r.remap_numpy(Dataset({'a': DataArray(
        np.zeros((0, m.n_a)), dims=('Time', 'nCells'))}), 0.1)
    assert out['a'].shape == (0, 30, 40)
    got = r.remap_array(np.zeros((0, m.n_a)), [1], 0.1)
    assert got.shape == (0, 30, 40)


@pytest.mark.parametrize('grid', [(50, 80), (4000,)], ids=['2d', '1d'])
def test_short_level_runs_take_small_lds_patches(dev, grid):
    """`(Time, nCells, L)` with 4 <= L < 16 on a row-group mapping: the plan
    builds small LDS patches on first use and the patch kernel (family 5)
    serves 7 <= L < 16, the batch-at-a-time lanes-across-rows kernel (family
    7, RUNS) on 256-row patches serves 4 <= L <= 6; bitwise, every mode and
    dtype; longer runs and partial row ranges keep the row groups."""
    from oracle import oracle
    from pyremap_amd import engine
    m, plan, csr, frac_b, choice = _problem(dev, grid)
    assert choice['family'] == 'rowgroup'
    rng = np.random.default_rng(11)
    assert plan._runs is None and plan._run_cells is None
    for shape in ((20, m.n_a, 4), (9, m.n_a, 10), (7, m.n_a, 12),
                  (5, m.n_a, 15), (2, 9, m.n_a, 5), (13, m.n_a, 7),
                  (11, m.n_a, 6), (1, 17, m.n_a, 4)):
        axes = [len(shape) - 2]
        for dtype in (np.float64, np.float32):
            x = rng.standard_normal(shape).astype(dtype)
            x[..., rng.random(m.n_a) < 0.15, :] = np.nan
            for thr in (None, 0.1):
                arg = x if thr is None else np.ma.masked_array(x, np.isnan(x))
                ref = np.ma.filled(oracle.remap_numpy_array(
                    csr, frac_b, grid, arg, axes, thr), np.nan)
                y = engine.remap_tensor(
                    plan, grid, torch.from_numpy(x).to(dev), axes,
                    engine.MODE_FRACB if thr is None else engine.MODE_MASKED,
                    threshold=thr or 0.0)
                assert_bitwise(y.cpu().numpy(), ref,
                               f'{shape} {dtype.__name__} {thr}')
        if shape == (20, m.n_a, 4):
            assert plan._run_cells and plan._runs is None
    assert plan._runs and plan._runs['rows'] <= 32
    assert plan._run_cells and 32 < plan._run_cells['rows'] <= 256
    # 16 levels and more: the row groups, no patch plan needed
    m2, plan2, csr2, frac2, _ = _problem(dev, grid, seed=4)
    x = rng.standard_normal((4, m2.n_a, 16))
    y = engine.remap_tensor(plan2, grid, torch.from_numpy(x).to(dev), [1],
                            engine.MODE_FRACB)
    assert plan2._runs is None and plan2._run_cells is None
    ref = np.ma.filled(oracle.remap_numpy_array(csr2, frac2, grid, x, [1],
                                                None), np.nan)
    assert_bitwise(y.cpu().numpy(), ref, '16 levels')


@pytest.mark.parametrize('dims,rows,block', [((256, 300), 512, 512),
                                             ((300, 480), 1024, 1024)])
def test_large_patches_persistent_workgroups(dev, dims, rows, block):
    """
    Grids of >= 64 K / 128 K cells take 16 x 32 / 32 x 32 patches: one 512- /
    1 024-thread workgroup per patch walks all its chunks (spmm_patchtime: a
    lane per row, two cells per lane, two LDS images) -- config 3's
    (120, nCells) runs this way.  Every value the oracle's, three modes, both
    dtypes, a K that is no multiple of the fields per lane, (Time, n, 2).
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    n_a = int(0.9 * dims[0] * dims[1])
    m = synthetic.conservative_map(n_a, dims, 3, 7, seed=21, device=dev,
                                   locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    plan.auto_schedule(m.dst_dims)
    q = plan.cell_patches()
    assert q['rows'] == rows and q['umax'] <= 2 * block
    rowptr, c, v = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, c, v, (m.n_b, m.n_a))
    frac_b = plan.frac_b.cpu().numpy()
    rng = np.random.default_rng(5)
    for shape, axes in (((9, n_a), [1]), ((3, n_a, 2), [1])):
        for dtype in (np.float64, np.float32):
            x = rng.standard_normal(shape).astype(dtype)
            holes = x.copy()
            holes[(slice(None),) * axes[0] + (rng.random(n_a) < 0.2,)] = \
                np.nan
            for field, thr in ((x, None), (holes, 0.3), (holes, None)):
                masked = thr is not None
                arg = np.ma.masked_array(field, np.isnan(field)) if masked \
                    else field
                want = oracle.remap_numpy_array(csr, frac_b, m.dst_dims, arg,
                                                axes, thr, nthreads=8)
                got = engine.remap_tensor(
                    plan, m.dst_dims, torch.from_numpy(field).to(dev), axes,
                    engine.MODE_MASKED if masked else engine.MODE_FRACB,
                    threshold=thr or 0.0)
                assert_bitwise(got.cpu().numpy(), np.ma.filled(want, np.nan),
                               f'{dims} {shape} {dtype.__name__} thr {thr}')


def test_plan_handle_large_grid_takes_the_large_patches(dev):
    """The opaque C plan handle on a grid of >= 128 K cells: its short-run
    patch plan holds 32 x 32 tiles too (`remap_plan_prepare_short_runs`), a
    (Time, nCells) field goes through the 1 024-thread persistent
    workgroups -- the bits of the Python layer's launch on the same map."""
    import ctypes
    from pyremap_amd import engine, synthetic
    dims_t = (300, 480)
    n_a = int(0.9 * dims_t[0] * dims_t[1])
    m = synthetic.conservative_map(n_a, dims_t, 3, 7, seed=21, device=dev,
                                   locality='mesh')
    mm = m.numpy()
    lib = engine.load_library()
    handle = ctypes.c_void_p()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dims = (ctypes.c_int64 * 2)(*m.dst_dims)

    def host(a, t):
        return np.ascontiguousarray(a, dtype=t).ctypes.data

    assert lib.remap_plan_create(
        m.n_b, m.n_a, len(mm['S']), host(mm['row'], np.int32),
        host(mm['col'], np.int32), host(mm['S'], np.float64), 1,
        host(mm['frac_b'], np.float64), 1, dims, 2, stream,
        ctypes.byref(handle)) == 0, lib.remap_last_error()
    try:
        assert lib.remap_plan_prepare_short_runs(handle, stream) == 0, \
            lib.remap_last_error()
        x = torch.randn((9, m.n_a), dtype=torch.float64, device=dev)
        y = torch.empty((9, m.n_b), dtype=torch.float64, device=dev)
        f = engine._Field()
        f.X, f.Y = x.data_ptr(), y.data_ptr()
        f.x_dtype, f.mode = engine.DTYPE_F64, engine.MODE_FRACB
        f.n_batch, f.k_inner = 9, 1
        f.x_row_stride, f.x_batch_stride = 1, m.n_a
        f.y_row_stride, f.y_batch_stride = 1, m.n_b
        assert lib.remap_plan_apply(handle, ctypes.byref(f), stream) == 0, \
            lib.remap_last_error()
        plan = engine.RemapPlan.from_triplets(
            m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, index_base=1,
            device=dev)
        plan.auto_schedule(m.dst_dims)
        assert plan.cell_patches()['rows'] == 1024
        want = engine.remap_tensor(plan, m.dst_dims, x, [1],
                                   engine.MODE_FRACB)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(y, nan=-2.0).reshape(-1),
                           torch.nan_to_num(want, nan=-2.0).reshape(-1))
    finally:
        lib.remap_plan_destroy(handle)


def test_plan_handle_short_level_runs(dev):
    """The opaque C plan handle on a row-group mapping, `(Time, nCells, L)`
    with 4 <= L < 16: once `remap_plan_prepare_short_runs` has run it takes
    the same routes as the Python layer (batch-at-a-time kernel up to 6
    levels, small LDS patches beyond) -- the oracle's bits, masked and not,
    before and after preparing; 16 levels keep the row groups."""
    import ctypes
    from oracle import oracle
    from pyremap_amd import engine
    grid = (50, 80)
    m, plan, csr, frac_b, choice = _problem(dev, grid)
    assert choice['family'] == 'rowgroup'
    mm = m.numpy()
    lib = engine.load_library()
    handle = ctypes.c_void_p()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dims = (ctypes.c_int64 * 2)(*grid)

    def host(a, t):
        return np.ascontiguousarray(a, dtype=t).ctypes.data

    assert lib.remap_plan_create(
        m.n_b, m.n_a, len(mm['S']), host(mm['row'], np.int32),
        host(mm['col'], np.int32), host(mm['S'], np.float64), 1,
        host(mm['frac_b'], np.float64), 1, dims, 2, stream,
        ctypes.byref(handle)) == 0, lib.remap_last_error()
    rng = np.random.default_rng(3)
    try:
        for prepared in (False, True):
            if prepared:
                assert lib.remap_plan_prepare_short_runs(handle, stream) == 0
            for T, L in ((20, 4), (13, 6), (11, 7), (9, 10), (5, 15),
                         (4, 16)):
                for dtype in (np.float64, np.float32):
                    x = rng.standard_normal((T, m.n_a, L)).astype(dtype)
                    x[:, rng.random(m.n_a) < 0.15, :] = np.nan
                    for masked in (False, True):
                        arg = np.ma.masked_array(x, np.isnan(x)) if masked \
                            else x
                        want = np.ma.filled(oracle.remap_numpy_array(
                            csr, frac_b, grid, arg, [1],
                            0.1 if masked else None), np.nan)
                        xd = torch.from_numpy(x).to(dev)
                        y = torch.empty((T, m.n_b, L), dtype=torch.float64,
                                        device=dev)
                        f = engine._Field()
                        f.X, f.Y = xd.data_ptr(), y.data_ptr()
                        f.x_dtype = engine.DTYPE_F32 if dtype == np.float32 \
                            else engine.DTYPE_F64
                        f.mode = engine.MODE_MASKED if masked \
                            else engine.MODE_FRACB
                        f.threshold = 0.1 if masked else 0.0
                        f.n_batch, f.k_inner = T, L
                        f.x_row_stride, f.x_batch_stride = L, m.n_a * L
                        f.y_row_stride, f.y_batch_stride = L, m.n_b * L
                        assert lib.remap_plan_apply(
                            handle, ctypes.byref(f), stream) == 0, \
                            lib.remap_last_error()
                        torch.cuda.synchronize()
                        assert_bitwise(
                            y.cpu().numpy().reshape(want.shape), want,
                            f'prepared {prepared} ({T}, n, {L}) '
                            f'{dtype.__name__} masked {masked}')
    finally:
        lib.remap_plan_destroy(handle)


@pytest.mark.parametrize('case', ['fine_to_coarse', 'coarse_to_fine',
                                  'coarse_to_fine_many'])
def test_plan_handle_and_python_agree_on_the_cell_patches(dev, case):
    """`remap_plan_prepare_short_runs` (C) and `RemapPlan.cell_patches`
    (Python) build the lanes-across-rows patch plan by ONE rule -- the tile
    halved until it fits; on a coarse -> fine map whose large patches are
    few, straight down to <= 256 rows, decided once: the same rows per patch
    on both sides (`remap_plan_info.cell_patch_rows`)."""
    import ctypes
    from pyremap_amd import engine, synthetic
    if case == 'fine_to_coarse':
        m = synthetic.conservative_map(120000, (300, 480), 3, 7, seed=2,
                                       device=dev, locality='mesh')
    else:
        # bilinear from a coarse grid: ~4 entries per row, few source cells
        # per patch (a 600-patch and a 2 400-patch case either side of the
        # "few large patches" rule)
        src = (150, 256) if case == 'coarse_to_fine' else (300, 512)
        dst = (src[0] * 4, src[1] * 4)
        m = synthetic.bilinear_map(src, dst, device=dev)
    mm = m.numpy()
    lib = engine.load_library()
    handle = ctypes.c_void_p()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dims = (ctypes.c_int64 * 2)(*m.dst_dims)

    def host(a, t):
        return np.ascontiguousarray(a, dtype=t).ctypes.data

    assert lib.remap_plan_create(
        m.n_b, m.n_a, len(mm['S']), host(mm['row'], np.int32),
        host(mm['col'], np.int32), host(mm['S'], np.float64), 1,
        host(mm['frac_b'], np.float64), 1, dims, 2, stream,
        ctypes.byref(handle)) == 0, lib.remap_last_error()
    try:
        assert lib.remap_plan_prepare_short_runs(handle, stream) == 0, \
            lib.remap_last_error()
        info = engine._PlanInfo()
        assert lib.remap_plan_query(handle, ctypes.byref(info)) == 0
        plan = engine.RemapPlan.from_triplets(
            m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, index_base=1,
            device=dev)
        plan.auto_schedule(m.dst_dims)
        q = plan.cell_patches()
        assert info.cell_patch_rows == q['rows'], \
            (case, info.cell_patch_rows, q['rows'], q['tile'], q['umax'])
    finally:
        lib.remap_plan_destroy(handle)
