"""
Round 6: masks that do not change from batch to batch.

* `spmm_grouptime` (csrc/spmm_grouptime.h): the masked mode of the 8-row
  groups under REMAP_FLAG_BATCH_MASKS -- lanes across the levels, four time
  slices per lane, ONE normaliser per lane and row while the four slices of a
  lane are valid or missing together; a group that meets anything else is
  redone with per-element normalisers inside the launch.  Every value against
  the oracle, bit for bit, whatever is missing and whether or not the flag is
  passed (reference: remap_numpy.py:262-266, 277-278).
* `remap_scan_nan_layout`: the NaN scan told where the field's cells and
  batches are, and the form of the masked launch it names; the four gated
  launches of `remap_tensor_auto_mode` / `remap_plan_apply_auto`.
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


@pytest.fixture(scope='module')
def problem(dev):
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(1500, (38, 60), 6, 22, seed=5,
                                   signed=True, locality='mesh')
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(
        mm['row'], mm['col'], mm['S'], mm['frac_b'], m.n_a, m.n_b,
        index_base=1, device=dev)
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    choice = plan.auto_schedule(m.dst_dims)
    assert choice['rows_per_group'] == 8
    return m, mm, plan, csr


def _fields(n_a, T, L, seed):
    """(tag, (T, n_a, L) field, form the layout scan should name)."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((T, n_a, L))
    out = [('no NaN', base.copy(), 0)]
    x = base.copy()
    x[:, rng.random(n_a) < 0.25, :] = np.nan          # land: whole cells
    x[:, 0, :] = np.nan
    out.append(('whole cells', x, 1))
    x = base.copy()
    depth = rng.integers(1, L + 1, n_a)                # bathymetry
    x[:, np.arange(L)[None, :] >= depth[:, None]] = np.nan
    out.append(('bathymetry', x, 2 if T >= 3 else 3))
    x = x.copy()
    x[:, rng.random(n_a) < 0.2, :] = np.nan            # + land
    out.append(('bathymetry and land', x, 2 if T >= 3 else 3))
    x = base.copy()
    x[rng.random(x.shape) < 0.02] = np.nan             # changes with time
    x[-1, -1, -1] = np.nan
    out.append(('single values', x, 3))
    x = base.copy()
    x[:, np.arange(L)[None, :] >= depth[:, None]] = np.nan
    x[T // 2, 7, 0] = np.nan                           # one slice differs
    out.append(('bathymetry but one value', x, 3))
    return out


def _reference(csr, frac_b, f, thr, masked=None):
    from oracle import oracle
    T, n_a, L = f.shape
    flat = np.ascontiguousarray(f.transpose(1, 0, 2)).reshape(n_a, T * L)
    if masked is None:     # the reference's branch (remap_numpy.py:201-204)
        masked = bool(np.isnan(flat).any())
    ref, ref_mask = oracle.remap_flat(csr, frac_b, flat.astype(np.float64),
                                      masked, thr)
    ref = ref.copy()
    ref[ref_mask] = np.nan
    n_b = ref.shape[0]
    return ref.reshape(n_b, T, L).transpose(1, 0, 2)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('shape', [(8, 64), (5, 60), (3, 100), (16, 7),
                                   (4, 130), (2, 64)])
def test_batch_masks_form_bitwise(dev, problem, shape, dtype):
    """REMAP_FLAG_BATCH_MASKS on every kind of field, with and without the
    other hint, float32 fields, level runs shorter / longer than a wave,
    time-block tails; two batches: the flag is ignored."""
    from pyremap_amd import engine
    m, mm, plan, csr = problem
    T, L = shape
    for tag, f, _ in _fields(m.n_a, T, L, T * 131 + L):
        f = f.astype(dtype)
        ref = _reference(csr, mm['frac_b'], f, 0.3, masked=True)
        fd = torch.from_numpy(f).to(dev)
        for flags in (engine.FLAG_BATCH_MASKS, 0,
                      engine.FLAG_BATCH_MASKS | engine.FLAG_CELL_MASKS):
            # (the plan's own choice: through the LDS ring where the field
            # allows; tune[5] = 9: the form without the ring)
            for tune in (None, [10, 1, 1, 1, 3], [10, 4, 1, 2, 2],
                         [10, 1, 1, 1, 3, 9]):
                y, mask = engine.remap_tensor(
                    plan, None, fd, [1], engine.MODE_MASKED, threshold=0.3,
                    flags=flags, tune=tune, want_mask=True)
                what = f'{tag} (T={T}, L={L}) flags={flags} tune={tune}'
                assert_bitwise(y.cpu().numpy(), ref, what)
                assert np.array_equal(mask.cpu().numpy().astype(bool),
                                      np.isnan(ref)), what


def test_layout_scan_names_the_form_and_the_gated_launches(dev, problem):
    """remap_scan_nan_layout on (Time, nCells, L) fields: land cells are
    whole cells (the flat scan of round 5 called them "column by column"),
    bathymetry is the same mask in every batch; remap_tensor_auto_mode
    enqueues the four gated launches and the one that runs gives the
    reference's result (remap_numpy.py:201-204, 258-278)."""
    from pyremap_amd import engine
    m, mm, plan, csr = problem
    for T, L in ((6, 64), (4, 60), (2, 64), (1, 192)):
        for tag, f, form in _fields(m.n_a, T, L, T + L):
            fd = torch.from_numpy(f).to(dev)
            kinds = torch.zeros(4, dtype=torch.int32, device=dev)
            engine.scan_nan_layout(fd, m.n_a, T, L, kinds)
            got = kinds.tolist()
            has = bool(np.isnan(f).any())
            cells = np.isnan(f).all(axis=(0, 2)) | \
                ~np.isnan(f).any(axis=(0, 2))
            same = (np.isnan(f) == np.isnan(f[:1])).all()
            want = [int(has), 0 if not has else 1 if cells.all() else 3,
                    0 if not has else 1 if same else 3]
            want.append(0 if not has else 1 if cells.all() else
                        2 if same and T >= 3 else 3)
            assert got == want, (tag, T, L, got, want)
            if T > 1:
                assert got[3] == form, (tag, T, L, got)
            y = engine.remap_tensor_auto_mode(plan, m.dst_dims, fd, [1], 0.3)
            ref = _reference(csr, mm['frac_b'], f, 0.3)
            assert_bitwise(y.cpu().numpy().reshape(ref.shape), ref,
                           f'auto {tag} (T={T}, L={L})')
    # float32, and a field with nothing in it
    f32 = np.random.default_rng(1).standard_normal((3, m.n_a, 8)).astype(
        np.float32)
    f32[:, 5, :] = np.nan
    kinds = torch.zeros(4, dtype=torch.int32, device=dev)
    engine.scan_nan_layout(torch.from_numpy(f32).to(dev), m.n_a, 3, 8, kinds)
    assert kinds.tolist() == [1, 1, 1, 1]
    kinds.zero_()
    engine.scan_nan_layout(torch.zeros((0, m.n_a, 8), device=dev), m.n_a, 0,
                           8, kinds)
    assert kinds.tolist() == [0, 0, 0, 0]


def test_plan_handle_auto_with_four_forms(dev, problem):
    """remap_plan_apply_auto (the C plan handle): the layout scan and the
    gated launches issued by the library; kinds comes back with the form."""
    import ctypes
    from pyremap_amd import engine
    m, mm, plan, csr = problem
    lib = engine.load_library()
    handle = ctypes.c_void_p()
    row = torch.from_numpy(mm['row'].astype(np.int32)).to(dev)
    col = torch.from_numpy(mm['col'].astype(np.int32)).to(dev)
    S = torch.from_numpy(mm['S']).to(dev)
    fb = torch.from_numpy(mm['frac_b']).to(dev)
    dims = (ctypes.c_int64 * 2)(*m.dst_dims)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    engine._check(lib.remap_plan_create(
        m.n_b, m.n_a, row.numel(), row.data_ptr(), col.data_ptr(),
        S.data_ptr(), 1, fb.data_ptr(), 1, dims, 2, stream,
        ctypes.byref(handle)), 'remap_plan_create')
    try:
        T, L = 6, 64
        for tag, f, form in _fields(m.n_a, T, L, 99):
            fd = torch.from_numpy(f).to(dev)
            y = torch.full((T, m.n_b, L), 7.0, dtype=torch.float64,
                           device=dev)
            kinds = torch.full((4,), 9, dtype=torch.int32, device=dev)
            fld = engine._Field()
            fld.X, fld.x_dtype = fd.data_ptr(), 0
            fld.n_batch, fld.k_inner = T, L
            fld.x_row_stride, fld.x_batch_stride = L, m.n_a * L
            fld.Y = y.data_ptr()
            fld.y_row_stride, fld.y_batch_stride = L, m.n_b * L
            fld.threshold = 0.3
            engine._check(lib.remap_plan_apply_auto(
                handle, ctypes.byref(fld), fd.numel(), kinds.data_ptr(),
                stream), 'remap_plan_apply_auto')
            assert kinds.tolist()[3] == form, (tag, kinds.tolist())
            ref = _reference(csr, mm['frac_b'], f, 0.3)
            assert_bitwise(y.cpu().numpy(), ref, f'plan handle, {tag}')
    finally:
        lib.remap_plan_destroy(handle)


def test_batches_further_apart_than_32_bit_offsets(dev, problem):
    """(Time, nCells, nVertLevels) on a 3.7 M-cell mesh: the time slices lie
    1.9 GB apart.  The forms of family 10 that address X through the LDS-DMA
    take 64-bit offsets -- frac_b (the shared form), masked with either hint
    -- and give the bits of the same call on a compact field; the other
    forms decline (under REMAP_FLAG_TUNE_HINT: the wave-per-row kernel with
    64-bit addressing)."""
    from pyremap_amd import engine
    m, mm, plan, csr = problem
    T, L = 5, 64
    stride = (1 << 28) + 64           # elements: 2 GiB and a bit per slice
    rng = np.random.default_rng(21)
    f = rng.standard_normal((T, m.n_a, L))
    depth = rng.integers(1, L + 1, m.n_a)
    fm = f.copy()
    fm[:, np.arange(L)[None, :] >= depth[:, None]] = np.nan
    fm[:, rng.random(m.n_a) < 0.2, :] = np.nan
    big = torch.zeros(T * stride, dtype=torch.float64, device=dev)
    y = torch.empty((T, m.n_b, L), dtype=torch.float64, device=dev)

    def run(field, mode, flags, tune):
        view = torch.as_strided(big, (T, m.n_a, L), (stride, L, 1))
        view.copy_(torch.from_numpy(field).to(dev))
        y.fill_(7.0)
        engine.apply_strided(
            plan, big, y, n_batch=T, k_inner=L, x_row_stride=L,
            x_batch_stride=stride, y_row_stride=L, y_batch_stride=m.n_b * L,
            mode=mode, threshold=0.3, flags=flags, tune=tune)
        return y.cpu().numpy()

    ref = _reference(csr, mm['frac_b'], f, 0.3, masked=False)
    assert_bitwise(run(f, engine.MODE_FRACB, 0, [10, 0, 2, 0, 3, 32]), ref,
                   'frac_b, shared form, wide batches')
    assert_bitwise(run(f, engine.MODE_FRACB, engine.FLAG_TUNE_HINT,
                       [10, 0, 2, 0, 3, 32]), ref, 'frac_b, hint')
    refm = _reference(csr, mm['frac_b'], fm, 0.3, masked=True)
    for flags in (engine.FLAG_BATCH_MASKS, engine.FLAG_CELL_MASKS):
        assert_bitwise(run(fm, engine.MODE_MASKED, flags, [10, 1, 1, 1, 3]),
                       refm, f'masked, flags {flags}, wide batches')
    # neither hint: the 8-row groups cannot reach; declined / handed on
    with pytest.raises(engine.EngineError, match='32-bit offsets'):
        run(fm, engine.MODE_MASKED, 0, [10, 1, 1, 1, 3])
    assert_bitwise(run(fm, engine.MODE_MASKED, engine.FLAG_TUNE_HINT,
                       [10, 1, 1, 1, 3]), refm, 'masked, hint, wide batches')
    # a mask that changes with time: the time form redoes its groups
    fv = fm.copy()
    fv[2, 11, 3] = np.nan
    fv[4, 200:260, :] = np.nan
    refv = _reference(csr, mm['frac_b'], fv, 0.3, masked=True)
    assert_bitwise(run(fv, engine.MODE_MASKED, engine.FLAG_BATCH_MASKS,
                       [10, 1, 1, 1, 3]), refv, 'varying mask, wide batches')


def test_host_array_column_panels(dev, problem):
    """host_path._panel_pipeline: an (n_a, K) host array goes up, is remapped
    and comes down in column panels (strided 2-D copies, both PCIe
    directions busy) -- the reference's result (remap_numpy.py:254-278) in
    every mode, for float32 input, a last panel that is not full, masks, and
    mode 'auto' on a field whose only NaN the strided sample does not meet
    (the device scans find it: the masked mode over again)."""
    from oracle import oracle
    from pyremap_amd import engine, host_path
    m, mm, plan, csr = problem
    rng = np.random.default_rng(31)
    old = host_path.CHUNK_BYTES, host_path.PANEL_COLUMNS
    host_path.CHUNK_BYTES = 1 << 16       # these small fields qualify
    calls = []
    inner = host_path._panel_pipeline

    def spy(*a, **k):
        out = inner(*a, **k)
        calls.append(out is not None)
        return out
    host_path._panel_pipeline = spy
    try:
        for K, kp in ((512, 128), (450, 128), (384, 64)):
            host_path.PANEL_COLUMNS = kp
            x = rng.standard_normal((m.n_a, K))
            holed = x.copy()
            holed[rng.random(m.n_a) < 0.2, K // 3:] = np.nan
            rare = x.copy()
            rare[m.n_a // 2 + 1, K - 3] = np.nan      # one NaN, off the sample
            assert not host_path._sampled_nan(rare)
            for field, mode, thr in ((x, 'fracb', None),
                                     (holed, 'masked', 0.3),
                                     (holed, 'auto', 0.3), (x, 'auto', 0.3),
                                     (rare, 'auto', 0.3),
                                     (x.astype(np.float32), 'fracb', None)):
                masked = mode == 'masked' or (mode == 'auto' and
                                              np.isnan(field).any())
                arg = np.ma.masked_array(field, np.isnan(field)) if masked \
                    else field
                ref = oracle.remap_numpy_array(
                    csr, mm['frac_b'], m.dst_dims, arg, [0],
                    thr if masked else None)
                want_mask = mode != 'auto'
                n0 = len(calls)
                got = host_path.remap_host_array(
                    plan, m.dst_dims, field, [0], mode=mode, threshold=thr,
                    want_mask=want_mask).result()
                # (a NaN the sample meets decides 'auto' before the upload:
                # masked, still in panels)
                assert calls[n0:] == [True], (K, kp, mode, calls[n0:])
                data = got[0] if want_mask else got
                what = f'panels K={K} kp={kp} {mode} {field.dtype}'
                assert_bitwise(data, np.ma.filled(ref, np.nan), what)
                if want_mask:
                    assert np.array_equal(got[1], np.ma.getmaskarray(ref)), \
                        what
        # too few panels: declined, the plain form answers
        host_path.PANEL_COLUMNS = 128
        x = rng.standard_normal((m.n_a, 200))
        n0 = len(calls)
        got = host_path.remap_host_array(plan, m.dst_dims, x, [0],
                                         mode='fracb').result()
        assert calls[n0:] == [False]
        ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, x, [0],
                                       None)
        assert_bitwise(got, np.ma.filled(ref, np.nan), 'plain form')
    finally:
        host_path._panel_pipeline = inner
        host_path.CHUNK_BYTES, host_path.PANEL_COLUMNS = old


@pytest.mark.parametrize('name', ['config3', 'headline', 'config4',
                                  'config5'])
def test_csr_from_coo_at_baseline_scale_against_the_oracle(dev, name):
    """`remap_csr_from_coo` (the device COO -> CSR of `_load_mapping`,
    remap_numpy.py:134-137: stable sort by (row, col), duplicates summed in
    input order) on BASELINE's own mappings -- config 3's 912 860 triplets
    (45 625 of them duplicates that merge), the headline's 8 million, config
    4's 110 million and config 5's 95 million (16.5 million duplicates) --
    against the oracle's C restatement of scipy's coo -> csr: row pointers and
    column indices equal, every weight bit for bit.  (The full-size parity
    tests hand the oracle the DEVICE-built CSR; this is the test that the
    CSR itself is the reference's; tools/csr_at_scale.py prints the line
    recorded in profiles/r06_analysis/csr_at_scale.txt.)"""
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.make_config(name, device=dev, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=dev)
    rowptr, col, val = plan.to_host_csr()
    mm = m.numpy()
    assert mm['row'].size > plan.nnz or name == 'config4'   # duplicates merge
    ref = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    assert ref.nnz == plan.nnz
    assert np.array_equal(rowptr, ref.indptr)
    assert np.array_equal(col, ref.indices)
    assert np.array_equal(val.view(np.int64), ref.data.view(np.int64))


def test_masked_division_by_a_lane_normaliser_and_its_guards(dev):
    """`spmm_timeshare` divides the four time slices of a lane by ONE
    normaliser: for ordinary operands the reciprocal is refined once per lane
    and every element finished with a multiply and two FMAs -- the very
    instructions the full IEEE division sequence ends with when its scaling
    steps scale nothing (csrc/spmm_device.h: finish_row_lane_den); zeros,
    denormals, 1e-250, 1e300, Inf, a normaliser outside [2^-126, 2^126] send
    the wave's row the long way.  Every value equals the oracle's C division
    (`num[ok] /= den[ok]`, remap_numpy.py:277), bit for bit; a negative
    threshold keeps rows with a negative normaliser."""
    from oracle import oracle
    from pyremap_amd import engine
    rng = np.random.default_rng(12)
    n_cells, T, L = 32, 5, 64
    x = rng.standard_normal((T, n_cells, L)) * \
        10.0 ** rng.integers(-200, 150, (1, n_cells, 1))
    specials = [0.0, -0.0, 1e-310, -4e-320, 1e-250, 2.0 ** -800,
                np.nextafter(2.0 ** -800, 0.0), 2.0 ** 600,
                np.nextafter(2.0 ** 601, 1.0), 2.0 ** 601, 1e300, np.inf,
                -np.inf, 1.5e-241, 8.3e180]
    for j, s in enumerate(specials):            # cells 16 ... : one each,
        x[:, 16 + j, (37 * j) % L] = s          # in every time slice
    depth = rng.integers(1, L + 1, n_cells)     # bathymetry: same mask at
    depth[16:] = L                              # every time
    x[:, np.arange(L)[None, :] >= depth[:, None]] = np.nan
    dens = [1.0, 0.5, 0.3, 1.0 / 3.0, 1e-30, 1e-38, 2.0 ** -126,
            np.nextafter(2.0 ** -126, 0.0), 2.0 ** 126,
            np.nextafter(2.0 ** 127, 1.0), 1e38, 1e-310, 0.0, -0.5, 1e300,
            0.9999999999999999]
    row = np.arange(n_cells * len(dens))
    col = row % n_cells
    S = np.repeat(np.asarray(dens), n_cells)    # one entry per row: den = S
    n_b = row.size
    frac_b = np.ones(n_b)
    plan = engine.RemapPlan.from_triplets(row + 1, col + 1, S, frac_b,
                                          n_cells, n_b, index_base=1,
                                          device=dev)
    csr = oracle.coo_to_csr(row, col, S, n_b, n_cells)
    plan.build_groups(None, rows=8, share=4)
    fd = torch.from_numpy(x).to(dev)
    with np.errstate(all='ignore'):
        for thr in (0.0, 0.3, -1.0):
            ref = _reference(csr, frac_b, x, thr, masked=True)
            flat = np.ascontiguousarray(x.transpose(1, 0, 2)).reshape(
                n_cells, T * L)
            ref_mask = oracle.remap_flat(csr, frac_b, flat, True, thr)[1]
            ref_mask = ref_mask.reshape(n_b, T, L).transpose(1, 0, 2)
            for flags in (engine.FLAG_BATCH_MASKS, 0):
                y, mask = engine.remap_tensor(
                    plan, None, fd, [1], engine.MODE_MASKED, threshold=thr,
                    flags=flags, tune=[10, 1, 1, 1, 3], want_mask=True)
                what = f'division by a lane normaliser, thr={thr} ' \
                       f'flags={flags}'
                assert_bitwise(y.cpu().numpy(), ref, what)
                assert np.array_equal(mask.cpu().numpy().astype(bool),
                                      ref_mask), what
