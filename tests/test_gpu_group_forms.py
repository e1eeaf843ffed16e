"""
The round-5 forms of kernel family 10 (row groups), every value against the
oracle, bit for bit, through the C ABI:

* `spmm_groupmask` (csrc/spmm_groupmask.h): the masked mode of the 8-row
  groups with per-ROW normalisers while a source cell is valid or missing in
  all of a wave's columns, and the general form for the groups where it is
  not -- whole cells missing, single values missing, nothing missing, NaN /
  Inf weights;
* the chunk-minor work list (tune[4] = 3);
* the rolling form (csrc/spmm_grouproll.h, tune[5] = 26 / 28);
* 16-row groups (2 x 8 tiles).
Reference arithmetic: remap_numpy.py:258-278.
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


def _problem(dev, rows, n_a=1500, dims=(38, 60), k=(6, 22), seed=5):
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(n_a, dims, k[0], k[1], seed=seed,
                                   signed=True, locality='mesh')
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(
        mm['row'], mm['col'], mm['S'], mm['frac_b'], m.n_a, m.n_b,
        index_base=1, device=dev)
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    plan.build_groups(m.dst_dims, super_tile=32, rows=rows)
    return m, mm, plan, csr


def _check(plan, csr, frac_b, x, dev, mode, thr, tune, what, flags=0):
    from oracle import oracle
    from pyremap_amd import engine
    xd = torch.from_numpy(x).to(dev)
    y, mask = engine.remap_tensor(plan, None, xd, [0], mode, threshold=thr,
                                  tune=tune, want_mask=True, flags=flags)
    if mode == engine.MODE_RAW:
        assert_bitwise(y.cpu().numpy(),
                       oracle.csr_matvecs(csr, x.astype(np.float64)), what)
        return
    ref, ref_mask = oracle.remap_flat(csr, frac_b, x.astype(np.float64),
                                      mode == engine.MODE_MASKED, thr)
    ref = ref.copy()
    ref[ref_mask] = np.nan
    assert_bitwise(y.cpu().numpy(), ref, what)
    assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask), what


def _fields(n_a, K, seed):
    """(tag, field): nothing missing / whole cells / single values / both,
    the first and the last cell and column among them."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n_a, K))
    out = [('no NaN', base.copy())]
    x = base.copy()
    x[rng.random(n_a) < 0.25] = np.nan
    x[0] = np.nan
    out.append(('whole cells', x))
    x = base.copy()
    x[rng.random((n_a, K)) < 0.02] = np.nan
    x[-1, -1] = np.nan
    out.append(('single values', x))
    x = base.copy()
    x[rng.random(n_a) < 0.2] = np.nan
    x[rng.random(n_a) < 0.05, K // 2:] = np.nan     # deep levels only
    x[5, 0] = np.nan
    out.append(('cells and levels', x))
    return out


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('K', [130, 256, 300, 1024])
def test_masked_groups_per_row_normaliser(dev, K, dtype):
    """REMAP_FLAG_CELL_MASKS on 8-row groups -> spmm_groupmask, whatever is
    missing; the same bits as without the flag; float32 fields too (read as
    float32, summed in float64: scipy's upcast)."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, 8)
    for tag, x in _fields(m.n_a, K, K):
        x = x.astype(dtype)
        for tune in ([10, 1, 2, 1], [10, 4, 2, 2, 3], [10, 1, 1, 1],
                     [10, 1, 2, 1, 0, 9], None):
            for flags in (engine.FLAG_CELL_MASKS, 0,
                          engine.FLAG_CELL_MASKS | engine.FLAG_FMA):
                if flags & engine.FLAG_FMA:
                    continue      # (covered by test_cell_masks_fma_is_close)
                _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_MASKED,
                       0.3, tune, f'{tag} K={K} tune={tune} flags={flags}',
                       flags=flags)


def test_cell_masks_fma_is_close(dev):
    """REMAP_FLAG_FMA with REMAP_FLAG_CELL_MASKS: rtol 1e-13, same mask."""
    from oracle import oracle
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, 8)
    for tag, x in _fields(m.n_a, 512, 77):
        xd = torch.from_numpy(x).to(dev)
        y, mask = engine.remap_tensor(
            plan, None, xd, [0], engine.MODE_MASKED, threshold=0.3,
            want_mask=True,
            flags=engine.FLAG_CELL_MASKS | engine.FLAG_FMA)
        ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, True, 0.3)
        got = y.cpu().numpy()
        assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask), tag
        ok = ~ref_mask
        np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-12, atol=1e-13)


def test_scan_kinds_and_three_gated_launches(dev):
    """remap_scan_nan_kinds tells whole missing cells from values missing
    column by column; `remap_tensor_auto_mode` on an entry-rich mapping
    enqueues the masked branch in both forms and the frac_b branch, each
    gated -- the reference's result (remap_numpy.py:201-204, 258-278) for
    every kind of field."""
    from oracle import oracle
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, 8)
    assert engine.cell_mask_form(plan)
    want = {'no NaN': (0, 0), 'whole cells': (1, 1),
            'single values': (1, 3), 'cells and levels': (1, 3)}
    for tag, x in _fields(m.n_a, 384, 3):
        xd = torch.from_numpy(x).to(dev)
        kinds = torch.zeros(2, dtype=torch.int32, device=dev)
        engine.scan_nan(xd, kinds)
        assert tuple(kinds.tolist()) == want[tag], (tag, kinds.tolist())
        one = torch.zeros(1, dtype=torch.int32, device=dev)
        engine.scan_nan(xd, one)
        assert int(one) == want[tag][0]
        y = engine.remap_tensor_auto_mode(plan, m.dst_dims, xd, [0], 0.3)
        masked = bool(np.isnan(x).any())
        ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, masked, 0.3)
        ref = ref.copy()
        ref[ref_mask] = np.nan
        assert_bitwise(y.cpu().numpy().reshape(m.n_b, -1), ref, tag)
    # float32 fields: runs of 256 elements; a view that starts inside an
    # allocation has its first elements peeled (judged "column by column")
    x32 = np.random.default_rng(0).standard_normal((64, 256)).astype(
        np.float32)
    x32[3] = np.nan
    kinds = torch.zeros(2, dtype=torch.int32, device=dev)
    engine.scan_nan(torch.from_numpy(x32).to(dev), kinds)
    assert tuple(kinds.tolist()) == (1, 1)
    big = torch.from_numpy(np.concatenate([[0.0], x32.ravel().astype(
        np.float64)])).to(dev)
    kinds.zero_()
    engine.scan_nan(big[1:], kinds)
    assert tuple(kinds.tolist()) == (1, 3)


def test_masked_groups_odd_weights_take_the_general_form(dev):
    """A NaN or an Inf weight on a cell missing in every column: `a * 0.0` is
    NaN there, so the skip of the fast form does not apply."""
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(900, (24, 40), 6, 18, seed=9, signed=True,
                                   locality='mesh')
    mm = m.numpy()
    S = mm['S'].copy()
    S[7] = np.inf
    S[1000] = np.nan
    S[2000] = -np.inf
    plan = engine.RemapPlan.from_triplets(
        mm['row'], mm['col'], S, mm['frac_b'], m.n_a, m.n_b, index_base=1,
        device=dev)
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, S, m.n_b, m.n_a)
    plan.build_groups(m.dst_dims, super_tile=32, rows=8)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((m.n_a, 256))
    x[rng.random(m.n_a) < 0.3] = np.nan
    for c in (mm['col'][7], mm['col'][1000], mm['col'][2000]):
        x[c - 1] = np.nan
    for flags in (engine.FLAG_CELL_MASKS, 0):
        _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_MASKED, 0.1,
               [10, 1, 2, 1], f'odd weights, flags {flags}', flags=flags)


@pytest.mark.parametrize('rows', [4, 8, 16])
@pytest.mark.parametrize('K', [130, 512])
def test_group_forms_bitwise(dev, rows, K):
    """Rolling form, chunk-minor list, 16-row groups: three modes each."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, rows)
    rng = np.random.default_rng(K + rows)
    x = rng.standard_normal((m.n_a, K))
    x[rng.random(m.n_a) < 0.15] = np.nan
    x[rng.random((m.n_a, K)) < 0.01] = np.nan
    tunes = [[10, 1, 2, 1], [10, 1, 1, 1, 3], [10, 4, 2, 2, 3]]
    tunes += [[10, 1, 2, 1, 0, 28], [10, 1, 1, 1, 3, 26],
              [10, 4, 2, 2, 0, 28], [10, 2, 1, 2, 0, 26]]
    if rows == 16:
        tunes.append([10, 1, 1, 1, 0, 4])
    for tune in tunes:
        for mode, thr in ((engine.MODE_FRACB, 0.0),
                          (engine.MODE_MASKED, 0.2), (engine.MODE_RAW, 0.0)):
            _check(plan, csr, mm['frac_b'], x, dev, mode, thr, tune,
                   f'rows={rows} K={K} tune={tune} mode={mode}')


# ---------------------------------------------------------------------------
# A Dataset's small variables of one shape: remapped together
# (host_path.remap_host_batch; reference: the per-variable loop of
# remap_numpy.py:42-55)
# ---------------------------------------------------------------------------
def test_dataset_small_variables_are_batched_bitwise(dev, monkeypatch):
    from oracle import oracle
    from pyremap_amd import DataArray, Dataset, Remapper, host_path, synthetic
    from pyremap_amd.remapper import remap_numpy as rn
    m = synthetic.conservative_map(900, (20, 36), 2, 7, seed=12,
                                   locality='mesh')
    mm = m.numpy()
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)

    class Desc:
        pass
    src, dst = Desc(), Desc()
    src.dims, src.dim_sizes = ['nCells'], [m.n_a]
    dst.dims, dst.dim_sizes = ['lat', 'lon'], list(m.dst_dims)
    dst.coords, dst.mesh_name = {}, 'toy'
    r = Remapper.from_triplets(mm['row'], mm['col'], mm['S'], mm['frac_b'],
                               src, dst, device=dev)
    rng = np.random.default_rng(4)
    ds = Dataset()
    fields = {}
    for v in range(9):
        x = rng.standard_normal((1, m.n_a))
        if v in (1, 2, 5, 8):
            x[:, rng.random(m.n_a) < 0.2] = np.nan
        fields[f'v{v}'] = x
        ds[f'v{v}'] = DataArray(x, dims=('Time', 'nCells'),
                                attrs={'units': f'u{v}'})
    x3 = rng.standard_normal((3, m.n_a, 4)).astype(np.float32)
    x3[1, 7, 2] = np.nan
    ds['deep_a'] = DataArray(x3, dims=('Time3', 'nCells', 'nLev'))
    ds['deep_b'] = DataArray(x3[::-1].copy(),
                             dims=('Time3', 'nCells', 'nLev'))
    ds['count'] = DataArray(rng.integers(0, 9, (1, m.n_a)),
                            dims=('Time', 'nCells'))
    ds['scalar'] = DataArray(np.arange(3.0), dims=('Time3',))
    calls = []
    real = host_path.remap_host_batch

    def spy(plan, dims, arrays, axes, **kw):
        calls.append(len(arrays))
        return real(plan, dims, arrays, axes, **kw)
    monkeypatch.setattr(host_path, 'remap_host_batch', spy)
    for thr in (0.05, None):
        calls.clear()
        out = r.remap_numpy(ds, thr)
        # (Time, nCells): the nine float64 fields and the integer one (it is
        # upcast as scipy upcasts it) together; the two float32 3-D fields too
        assert sorted(calls) == [2, 10], calls
        assert list(out.data_vars) == list(ds.data_vars)
        for name in ds.data_vars:
            da = ds[name]
            got = out[name]
            if name == 'scalar':
                assert np.array_equal(got.values, da.values)
                continue
            assert got.attrs == da.attrs
            # the same variable on its own: the per-variable pipeline
            alone = rn._remap_data_array(da, r, thr)
            assert list(got.dims) == list(alone.dims)
            assert_bitwise(got.values, alone.values, f'{name} thr={thr}')
            x = np.asarray(da.values, dtype=np.float64)
            axis = da.dims.index('nCells')
            flat = np.moveaxis(x, axis, 0).reshape(m.n_a, -1)
            masked = thr is not None and bool(np.isnan(flat).any())
            ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], flat,
                                              masked, thr or 0.0)
            ref = ref.copy()
            ref[ref_mask] = np.nan
            want = np.moveaxis(ref.reshape((m.n_b,) + tuple(
                s for a, s in enumerate(x.shape) if a != axis)), 0, axis)
            want = want.reshape(got.values.shape)
            assert_bitwise(got.values, want, f'{name} thr={thr} vs oracle')
    # call after call (side streams, the feeder thread, the pinned result
    # buffers are shared between calls): the same bits every time, also with
    # earlier results still alive
    keep = [r.remap_numpy(ds, 0.05) for _ in range(4)]
    for other in keep[1:]:
        for name in ds.data_vars:
            assert_bitwise(np.asarray(other[name].values, dtype=np.float64),
                           np.asarray(keep[0][name].values,
                                      dtype=np.float64), f'repeat {name}')


@pytest.mark.parametrize('rich', [True, False])
def test_plan_handle_apply_auto(dev, rich):
    """`remap_plan_apply_auto`: `_remap_data_array`'s NaN branch and
    `_remap_numpy_array` in ONE C call (scan + gated launches; three of them
    on an entry-rich mapping) -- the reference's result for every kind of
    field (remap_numpy.py:201-204, 258-278)."""
    import ctypes
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(1500, (38, 60), *((6, 22) if rich
                                                     else (1, 6)),
                                   seed=5, signed=rich, locality='mesh')
    mm = m.numpy()
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
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
        info = engine._PlanInfo()
        assert lib.remap_plan_query(handle, ctypes.byref(info)) == 0
        assert (info.group_rows == 8) == rich
        K = 384
        kinds = torch.full((4,), 7, dtype=torch.int32, device=dev)
        for tag, x in _fields(m.n_a, K, 11):
            xd = torch.from_numpy(x).to(dev)
            y = torch.full((m.n_b, K), 5.0, dtype=torch.float64, device=dev)
            f = engine._Field()
            f.X, f.Y = xd.data_ptr(), y.data_ptr()
            f.x_dtype, f.mode = engine.DTYPE_F64, 99      # (mode is ignored)
            f.n_batch, f.k_inner = 1, K
            f.x_row_stride, f.x_batch_stride = K, 0
            f.y_row_stride, f.y_batch_stride = K, 0
            f.threshold = 0.3
            assert lib.remap_plan_apply_auto(
                handle, ctypes.byref(f), xd.numel(), kinds.data_ptr(),
                stream) == 0, lib.remap_last_error()
            torch.cuda.synchronize()
            masked = bool(np.isnan(x).any())
            ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, masked,
                                              0.3)
            ref = ref.copy()
            ref[ref_mask] = np.nan
            assert_bitwise(y.cpu().numpy(), ref, f'{tag} rich={rich}')
            # (any NaN, whole cells 1 / not 3, one batch: its own mask 1,
            # the form the masked launch took)
            want = {'no NaN': (0, 0, 0, 0),
                    'whole cells': (1, 1, 1, 1)}.get(tag, (1, 3, 1, 3))
            assert tuple(kinds.tolist()) == want, (tag, kinds.tolist())
    finally:
        lib.remap_plan_destroy(handle)
