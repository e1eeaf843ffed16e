"""
GPU parity tests: the HIP path, called through the C ABI, against

* the golden vectors produced by the reference's own code
  (tests/golden/*.npz, see oracle/make_goldens.py), and
* the CPU oracle (oracle/) on seeded inputs.

Tolerance: NONE in the default mode -- fp64 results must be bit-identical
(NaN placement identical; NaN payloads are not compared).  The opt-in FMA
mode is checked at rtol 1e-13.
"""
import os

import numpy as np
import pytest

from helpers import assert_bitwise, golden_cases, golden_files, golden_map

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

FILES = golden_files()
IDS = [os.path.basename(f) for f in FILES]


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from pyremap_amd import engine
    engine.load_library()
    return torch.device('cuda', 0)


class _Desc:
    def __init__(self, dims, sizes):
        self.dims = list(dims)
        self.dim_sizes = [int(s) for s in sizes]
        self.coords = {}
        self.mesh_name = 'golden'


def _remapper_for(m, dev):
    from pyremap_amd import Remapper
    src = m['src_grid_dims'][::-1]
    dst = m['dst_grid_dims'][::-1]
    return Remapper.from_triplets(
        m['row'], m['col'], m['S'], m['frac_b'],
        _Desc([f's{i}' for i in range(len(src))], src),
        _Desc([f'd{i}' for i in range(len(dst))], dst), device=dev)


@pytest.mark.parametrize('path', FILES, ids=IDS)
def test_device_csr_matches_scipy_golden(path, dev):
    """remap_csr_from_coo == the reference's csr_matrix((S, (row, col)))."""
    from pyremap_amd import engine
    m = golden_map(path)
    plan = engine.RemapPlan.from_triplets(
        m['row'], m['col'], m['S'], m['frac_b'], int(m['n_a']),
        int(m['n_b']), index_base=1, device=dev)
    rowptr, col, val = plan.to_host_csr()
    assert np.array_equal(rowptr, m['csr_indptr'])
    assert np.array_equal(col, m['csr_indices'])
    assert_bitwise(val, m['csr_data'], 'csr data')


@pytest.mark.parametrize('path', FILES, ids=IDS)
def test_remap_numpy_array_matches_reference_golden(path, dev):
    """_remap_numpy_array on the GPU == the reference's, bit for bit."""
    m = golden_map(path)
    remapper = _remapper_for(m, dev)
    n = 0
    for i, arg, axes, thr, out, mask in golden_cases(path):
        res = remapper.remap_array(arg, axes, thr)
        assert isinstance(res, np.ma.MaskedArray)
        assert res.shape == out.shape, f'case {i}'
        assert np.array_equal(np.ma.getmaskarray(res), mask), f'case {i}'
        assert_bitwise(res.filled(np.nan), out, f'{path} case {i}')
        n += 1
    assert n > 0


@pytest.mark.parametrize('path', FILES[:3], ids=IDS[:3])
def test_device_tensor_in_device_tensor_out(path, dev):
    """A device tensor stays on the device; NaN marks the masked cells."""
    m = golden_map(path)
    remapper = _remapper_for(m, dev)
    for i, arg, axes, thr, out, mask in golden_cases(path):
        field = np.ma.getdata(arg)
        is_ma = isinstance(arg, np.ma.MaskedArray)
        if thr is not None and not is_ma and np.isnan(field).any():
            continue  # 'plain' goldens: the array API would pick masked mode
        if thr is not None and is_ma and not np.isnan(field).any():
            continue  # MaskedArray without NaN: no device equivalent
        x = torch.from_numpy(np.ascontiguousarray(field)).to(dev)
        y = remapper.remap_array(x, axes, thr)
        assert y.is_cuda and y.dtype == torch.float64
        assert_bitwise(y.cpu().numpy(), out, f'{path} case {i}')


def test_unstable_duplicates_limit(dev, golden_dir):
    """Same documented limit as the oracle: see test_oracle_golden.py."""
    from pyremap_amd import engine
    m = golden_map(os.path.join(golden_dir, 'gx_unstable_dups.npz'))
    plan = engine.RemapPlan.from_triplets(
        m['row'], m['col'], m['S'], m['frac_b'], int(m['n_a']),
        int(m['n_b']), index_base=1, device=dev)
    rowptr, col, val = plan.to_host_csr()
    assert np.array_equal(rowptr, m['csr_indptr'])
    assert np.array_equal(col, m['csr_indices'])
    np.testing.assert_allclose(val, m['csr_data'], rtol=1e-13, atol=2e-15)


# ---------------------------------------------------------------------------
# kernel variants against the oracle
# ---------------------------------------------------------------------------

def _random_problem(seed, n_a, n_b, lo, hi, long_rows=0):
    from oracle import oracle
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for i in range(n_b):
        if rng.random() < 0.1:
            continue
        k = int(rng.integers(lo, hi + 1))
        if i < long_rows:
            k = int(rng.integers(65, 200))
        c = rng.choice(n_a, size=min(k, n_a), replace=False)
        rows += [i] * len(c)
        cols += c.tolist()
        vals += (rng.standard_normal(len(c))).tolist()
    row = np.asarray(rows, dtype=np.int32)
    col = np.asarray(cols, dtype=np.int32)
    S = np.asarray(vals)
    frac_b = rng.random(n_b)
    frac_b[rng.random(n_b) < 0.1] = 0.0
    csr = oracle.coo_to_csr(row, col, S, n_b, n_a)
    return row, col, S, frac_b, csr


@pytest.fixture(scope='module')
def problem(dev):
    from pyremap_amd import engine
    n_a, n_b = 700, 531
    row, col, S, frac_b, csr = _random_problem(11, n_a, n_b, 1, 9,
                                               long_rows=5)
    plan = engine.RemapPlan.from_triplets(row, col, S, frac_b, n_a, n_b,
                                          index_base=0, device=dev)
    return dict(plan=plan, csr=csr, frac_b=frac_b, n_a=n_a, n_b=n_b)


KS = [1, 2, 3, 7, 31, 32, 33, 64, 65, 127, 128, 129, 130, 256, 300, 512, 514]


def _x(seed, n_a, K, dtype, nan_frac):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n_a, K)).astype(dtype)
    if nan_frac:
        x[rng.random((n_a, K)) < nan_frac] = np.nan
    return x


@pytest.mark.parametrize('K', KS)
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('mode', ['raw', 'fracb', 'masked'])
def test_modes_and_widths_bitwise(problem, dev, K, dtype, mode):
    from oracle import oracle
    from pyremap_amd import engine
    p = problem
    x = _x(K, p['n_a'], K, dtype, 0.2 if mode == 'masked' else 0.0)
    xd = torch.from_numpy(x).to(dev)
    y = torch.empty((p['n_b'], K), dtype=torch.float64, device=dev)
    mask = torch.empty((p['n_b'], K), dtype=torch.uint8, device=dev)
    emode = {'raw': engine.MODE_RAW, 'fracb': engine.MODE_FRACB,
             'masked': engine.MODE_MASKED}[mode]
    engine.apply_strided(p['plan'], xd, y, n_batch=1, k_inner=K,
                         x_row_stride=K, x_batch_stride=0, y_row_stride=K,
                         y_batch_stride=0, mode=emode, threshold=0.3,
                         mask_out=mask)
    torch.cuda.synchronize()
    x64 = x.astype(np.float64)
    if mode == 'raw':
        ref = oracle.csr_matvecs(p['csr'], x64)
        ref_mask = np.zeros_like(ref, dtype=bool)
    else:
        ref, ref_mask = oracle.remap_flat(p['csr'], p['frac_b'], x64,
                                          mode == 'masked', 0.3)
        ref[ref_mask] = np.nan
    assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask)
    assert_bitwise(y.cpu().numpy(), ref, f'K={K} {mode}')


TUNES = [
    (1, 1, 1, 1, 1, 0), (1, 1, 1, 3, 2, 0), (1, 2, 1, 1, 1, 0),
    (1, 2, 1, 4, 2, 0), (1, 2, 2, 2, 2, 0), (1, 2, 4, 1, 2, 0),
    (1, 2, 4, 5, 1, 0), (2, 0, 0, 0, 0, 0),
    # scalar-cache metadata family
    (6, 1, 1, 1, 2, 0), (6, 1, 1, 5, 1, 0), (6, 2, 1, 4, 2, 0),
    (6, 2, 2, 3, 2, 0), (6, 2, 2, 8, 1, 0), (6, 0, 0, 0, 0, 0),
    # one element per lane, two 64-column tiles (odd strides take this)
    (6, 1, 2, 3, 2, 0), (6, 1, 0, 4, 1, 0),
]


@pytest.mark.parametrize('tune', TUNES)
@pytest.mark.parametrize('K', [6, 64, 200, 1024])
def test_launch_shapes_bitwise(problem, dev, tune, K):
    """Every kernel family / tile shape / block map gives the same bits."""
    from oracle import oracle
    from pyremap_amd import engine
    p = problem
    x = _x(100 + K, p['n_a'], K, np.float64, 0.1)
    xd = torch.from_numpy(x).to(dev)
    for emode, masked in ((engine.MODE_FRACB, False),
                          (engine.MODE_MASKED, True)):
        y = torch.full((p['n_b'], K), 7.0, dtype=torch.float64, device=dev)
        engine.apply_strided(p['plan'], xd, y, n_batch=1, k_inner=K,
                             x_row_stride=K, x_batch_stride=0,
                             y_row_stride=K, y_batch_stride=0, mode=emode,
                             threshold=0.05, tune=tune)
        ref, ref_mask = oracle.remap_flat(p['csr'], p['frac_b'], x, masked,
                                          0.05)
        ref[ref_mask] = np.nan
        assert_bitwise(y.cpu().numpy(), ref, f'tune={tune} K={K}')


def test_row_range_and_shards(problem, dev):
    """Row shards (rebased CSR) and [row_begin, row_end) launches."""
    from oracle import oracle
    from pyremap_amd import engine
    p = problem
    K = 96
    x = _x(5, p['n_a'], K, np.float64, 0.0)
    xd = torch.from_numpy(x).to(dev)
    ref, ref_mask = oracle.remap_flat(p['csr'], p['frac_b'], x, False, 0.0)
    ref[ref_mask] = np.nan
    # partial launch leaves the other rows untouched
    y = torch.full((p['n_b'], K), -1.0, dtype=torch.float64, device=dev)
    engine.apply_strided(p['plan'], xd, y, n_batch=1, k_inner=K,
                         x_row_stride=K, x_batch_stride=0, y_row_stride=K,
                         y_batch_stride=0, mode=engine.MODE_FRACB,
                         row_begin=100, row_end=333)
    got = y.cpu().numpy()
    assert_bitwise(got[100:333], ref[100:333])
    assert (got[:100] == -1.0).all() and (got[333:] == -1.0).all()
    # shards: nnz-balanced contiguous row ranges that tile [0, n_b)
    for world in (2, 3, 8):
        bounds = p['plan'].shard_bounds(world)
        assert bounds[0] == 0 and bounds[-1] == p['n_b']
        assert all(a <= b for a, b in zip(bounds, bounds[1:]))
        parts = []
        for rank in range(world):
            sh = p['plan'].shard(rank, world)
            assert sh.row_offset == bounds[rank]
            ys = engine.remap_tensor(sh, None, xd, [0], engine.MODE_FRACB)
            assert ys.shape == (sh.n_b, K)
            parts.append(ys.cpu().numpy())
        assert_bitwise(np.concatenate(parts, axis=0), ref)


def test_layouts_bitwise(dev):
    """(T, n, L) in place, source axes last, two source axes, 4-D."""
    from oracle import oracle
    from pyremap_amd import engine
    n_b = 12 * 9
    row, col, S, frac_b, csr = _random_problem(3, 20 * 15, n_b, 1, 6)
    plan = engine.RemapPlan.from_triplets(row, col, S, frac_b, 300, n_b,
                                          index_base=0, device=dev)
    rng = np.random.default_rng(0)
    shapes = [
        ((4, 300, 70), [1]), ((4, 300, 3), [1]), ((300,), [0]),
        ((5, 6, 300), [2]), ((2, 20, 15, 66), [1, 2]), ((20, 15), [0, 1]),
        ((3, 20, 15), [1, 2]), ((2, 3, 300, 2, 35), [2]),
        ((20, 7, 15), [0, 2]),
        # level counts that do not divide a wave's 128 (64) columns: whole
        # batches per K tile (60, 50, 61), the flat cut (100, 24), and
        # batches left over in the last chunk
        ((5, 300, 60), [1]), ((3, 300, 61), [1]), ((7, 300, 24), [1]),
        ((5, 300, 100), [1]), ((5, 300, 50), [1]), ((2, 2, 300, 60), [2]),
        ((3, 300, 65), [1]), ((2, 300, 101), [1]), ((2, 300, 127), [1]),
    ]
    for schedule in (None, 4, 8):
        if schedule is None:
            plan.groups = None
            plan.set_row_order(None)
        else:
            plan.build_groups((12, 9), rows=schedule)
        for shape, axes in shapes:
            if schedule is not None and shape[-1] < 24:
                continue      # the few-fields kernel: covered above
            f = rng.standard_normal(shape)
            fn = f.copy()
            fn[rng.random(shape) < 0.15] = np.nan
            for field, mode, thr in ((f, engine.MODE_FRACB, None),
                                     (fn, engine.MODE_MASKED, 0.1),
                                     (f.astype(np.float32),
                                      engine.MODE_FRACB, None)):
                arg = field if thr is None else np.ma.masked_array(
                    field, np.isnan(field))
                ref = oracle.remap_numpy_array(csr, frac_b, (12, 9), arg,
                                               axes, thr).filled(np.nan)
                y = engine.remap_tensor(
                    plan, (12, 9), torch.from_numpy(field).to(dev), axes,
                    mode, threshold=thr or 0.0,
                    tune=None if schedule is None else [10])
                assert tuple(y.shape) == ref.shape, (shape, axes)
                assert y.is_contiguous()
                assert_bitwise(y.cpu().numpy(), ref,
                               f'{shape} {axes} {schedule} {field.dtype}')


def test_fma_flag_is_close_not_identical(problem, dev):
    from oracle import oracle
    from pyremap_amd import engine
    p = problem
    K = 256
    x = _x(9, p['n_a'], K, np.float64, 0.0)
    xd = torch.from_numpy(x).to(dev)
    y = torch.empty((p['n_b'], K), dtype=torch.float64, device=dev)
    engine.apply_strided(p['plan'], xd, y, n_batch=1, k_inner=K,
                         x_row_stride=K, x_batch_stride=0, y_row_stride=K,
                         y_batch_stride=0, mode=engine.MODE_RAW,
                         flags=engine.FLAG_FMA)
    ref = oracle.csr_matvecs(p['csr'], x)
    got = y.cpu().numpy()
    scale = np.abs(ref).max()
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-13 * scale)
    assert not np.array_equal(got, ref)


def test_edge_cases(dev):
    from pyremap_amd import engine
    # no triplets at all: every row empty
    plan = engine.RemapPlan.from_triplets(
        np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0),
        np.asarray([1.0, 0.0, 0.5]), 4, 3, index_base=1, device=dev)
    assert plan.nnz == 0
    x = torch.ones((4, 5), dtype=torch.float64, device=dev)
    y = engine.remap_tensor(plan, (3,), x, [0], engine.MODE_FRACB).cpu()
    assert (y[0] == 0).all() and torch.isnan(y[1]).all() and (y[2] == 0).all()
    y = engine.remap_tensor(plan, (3,), x, [0], engine.MODE_MASKED,
                            threshold=0.0).cpu()
    assert torch.isnan(y).all()          # den = 0 is not > 0
    # K = 0: nothing to do, shape still right
    y = engine.remap_tensor(plan, (3,), x[:, :0], [0], engine.MODE_FRACB)
    assert tuple(y.shape) == (3, 0)
    # out-of-range triplets are rejected
    with pytest.raises(ValueError, match='outside'):
        engine.RemapPlan.from_triplets(
            np.asarray([1, 9], np.int32), np.asarray([1, 1], np.int32),
            np.ones(2), np.ones(3), 4, 3, index_base=1, device=dev)
    # wrong source size
    with pytest.raises(ValueError, match='n_a'):
        engine.remap_tensor(plan, (3,), torch.ones((5, 2), device=dev,
                                                   dtype=torch.float64),
                            [0], engine.MODE_FRACB)
    # integer fields are upcast like scipy does
    xi = torch.arange(8, device=dev, dtype=torch.int32).reshape(4, 2)
    plan2 = engine.RemapPlan.from_triplets(
        np.asarray([1, 1, 2], np.int32), np.asarray([1, 4, 2], np.int32),
        np.asarray([0.5, 0.25, 2.0]), np.asarray([1.0, 1.0]), 4, 2,
        index_base=1, device=dev)
    y = engine.remap_tensor(plan2, (2,), xi, [0], engine.MODE_FRACB).cpu()
    assert y.tolist() == [[0.5 * 0 + 0.25 * 6, 0.5 * 1 + 0.25 * 7],
                          [4.0, 6.0]]


def test_dataset_level_matches_reference_golden(dev, golden_dir):
    """Remapper.remap_numpy on a Dataset == the reference's _remap_numpy."""
    import json
    import sys

    from pyremap_amd import DataArray, Dataset, Remapper
    g = np.load(os.path.join(golden_dir, 'g3_dataset.npz'))
    meta = json.loads(str(g['meta_json']))

    def build_input(prefix, rec):
        ds = Dataset(attrs=rec['attrs'])
        for v in rec['data_vars']:
            ds[v['name']] = DataArray(g[f'{prefix}{v["name"]}'],
                                      dims=v['dims'], attrs=v['attrs'])
        for v in rec['coords']:
            ds._set_coord(v['name'], DataArray(g[f'{prefix}{v["name"]}'],
                                               dims=v['dims'],
                                               attrs=v['attrs']))
        return ds

    dst_coords = {
        k: {'dims': v['dims'], 'data': g[f'dst_{k}'], 'attrs': v['attrs']}
        for k, v in meta['dst_coords'].items()}

    class Desc:
        pass
    src = Desc()
    src.dims, src.dim_sizes = ['nCells'], [int(g['n_a'])]
    dst = Desc()
    dst.dims = ['lat', 'lon']
    dst.dim_sizes = [len(g['dst_lat']), len(g['dst_lon'])]
    dst.coords, dst.mesh_name = dst_coords, 'toy_6x8'

    def remapper():
        return Remapper.from_triplets(g['row'], g['col'], g['S'],
                                      g['frac_b'], src, dst, device=dev)

    ds = build_input('in__', meta['input'])
    old = sys.argv
    sys.argv = meta['argv']
    try:
        for tag, thr in (('thr', 0.01), ('nothr', None)):
            out = remapper().remap_numpy(ds, thr)
            rec = meta[f'dataset_{tag}']
            assert list(out.data_vars) == [v['name']
                                           for v in rec['data_vars']]
            assert sorted(out.coords) == sorted(v['name']
                                                for v in rec['coords'])
            assert {k: str(v) for k, v in out.attrs.items()} == rec['attrs']
            for v in rec['data_vars'] + rec['coords']:
                var = out.variables[v['name']]
                assert list(var.dims) == v['dims'], v['name']
                assert str(var.dtype) == v['dtype'], v['name']
                assert {k: str(a) for k, a in var.attrs.items()} == \
                    v['attrs'], v['name']
                assert_bitwise(var.values.astype(np.float64),
                               g[f'{tag}__{v["name"]}'].astype(np.float64),
                               f'{tag} {v["name"]}')
        da = remapper().remap_numpy(ds['temperature'], 0.01)
        rec = meta['dataarray_thr']
        assert da.name == rec['name'] and list(da.dims) == rec['dims']
        assert sorted(da.coords) == sorted(c['name'] for c in rec['coords'])
        assert_bitwise(da.values, g['da__data'])
    finally:
        sys.argv = old

    # partial source dims are dropped from a Dataset (2-D source)
    src2 = Desc()
    src2.dims, src2.dim_sizes = ['y', 'x'], [5, 10]
    r2 = Remapper.from_triplets(g['p_map__row'], g['p_map__col'],
                                g['p_map__S'], g['p_map__frac_b'], src2, dst,
                                device=dev)
    ds2 = build_input('p_in__', meta['partial_input'])
    out2 = r2.remap_numpy(ds2, None)
    rec2 = meta['partial_dataset']
    assert list(out2.data_vars) == [v['name'] for v in rec2['data_vars']]
    for v in rec2['data_vars']:
        assert list(out2.variables[v['name']].dims) == v['dims']
        assert_bitwise(out2.variables[v['name']].values,
                       g[f'p_out__{v["name"]}'])


def test_dataset_level_with_real_xarray(dev, golden_dir):
    """
    G3 ran the reference with this package's `xr_lite` standing in for
    xarray on both sides (xarray is absent from the image).  The day xarray
    imports, the same inputs go through REAL `xarray.Dataset` /
    `DataArray` objects -- `Dataset.map(keep_attrs=True)`,
    `DataArray.from_dict`, remap_numpy.py:42-55, 209-218 -- and must give the
    committed outputs.  Skipped otherwise.
    """
    xr = pytest.importorskip('xarray')
    import json
    import sys

    from pyremap_amd import Remapper
    g = np.load(os.path.join(golden_dir, 'g3_dataset.npz'))
    meta = json.loads(str(g['meta_json']))
    rec_in = meta['input']
    ds = xr.Dataset(
        {v['name']: (v['dims'], g[f'in__{v["name"]}'], v['attrs'])
         for v in rec_in['data_vars']},
        coords={v['name']: (v['dims'], g[f'in__{v["name"]}'], v['attrs'])
                for v in rec_in['coords']},
        attrs=rec_in['attrs'])

    class Desc:
        pass
    src, dst = Desc(), Desc()
    src.dims, src.dim_sizes = ['nCells'], [int(g['n_a'])]
    dst.dims = ['lat', 'lon']
    dst.dim_sizes = [len(g['dst_lat']), len(g['dst_lon'])]
    dst.coords = {k: {'dims': v['dims'], 'data': g[f'dst_{k}'],
                      'attrs': v['attrs']}
                  for k, v in meta['dst_coords'].items()}
    dst.mesh_name = 'toy_6x8'
    old = sys.argv
    sys.argv = meta['argv']
    try:
        for tag, thr in (('thr', 0.01), ('nothr', None)):
            r = Remapper.from_triplets(g['row'], g['col'], g['S'],
                                       g['frac_b'], src, dst, device=dev)
            out = r.remap_numpy(ds, thr)
            assert isinstance(out, xr.Dataset)
            rec = meta[f'dataset_{tag}']
            assert list(out.data_vars) == [v['name']
                                           for v in rec['data_vars']]
            assert sorted(out.coords) == sorted(v['name']
                                                for v in rec['coords'])
            assert {k: str(v) for k, v in out.attrs.items()} == rec['attrs']
            for v in rec['data_vars'] + rec['coords']:
                var = out[v['name']]
                assert list(var.dims) == v['dims'], v['name']
                assert {k: str(a) for k, a in var.attrs.items()} == \
                    v['attrs'], v['name']
                assert_bitwise(np.asarray(var.values, dtype=np.float64),
                               g[f'{tag}__{v["name"]}'].astype(np.float64),
                               f'xarray {tag} {v["name"]}')
        da = r.remap_numpy(ds['temperature'], 0.01)
        assert isinstance(da, xr.DataArray)
        rec = meta['dataarray_thr']
        assert da.name == rec['name'] and list(da.dims) == rec['dims']
        assert_bitwise(np.asarray(da.values), g['da__data'])
    finally:
        sys.argv = old


# ---------------------------------------------------------------------------
# LDS-staged patch family
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('row_bytes', [1024, 512])
@pytest.mark.parametrize('tile', [(4, 8), (1, 16), (8, 8), (3, 5)])
@pytest.mark.parametrize('K', [64, 128, 130, 384, 512])
def test_patch_kernel_bitwise(dev, tile, K, row_bytes):
    """The LDS-staged schedule gives the same bits as the oracle."""
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(1500, (24, 40), 1, 7, seed=13)
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(mm['row'], mm['col'], mm['S'],
                                          mm['frac_b'], m.n_a, m.n_b,
                                          device=dev)
    ratio = plan.build_patches(m.dst_dims, tile=tile, row_bytes=row_bytes)
    assert ratio is not None and 0 < ratio <= 1
    assert plan.patches['row_bytes'] == row_bytes
    assert 1 <= plan.patches['rows'] <= tile[0] * tile[1]
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    rng = np.random.default_rng(K)
    x = rng.standard_normal((m.n_a, K))
    x[rng.random(m.n_a) < 0.2, :] = np.nan
    xd = torch.from_numpy(x).to(dev)
    for emode, masked in ((engine.MODE_FRACB, False),
                          (engine.MODE_MASKED, True),
                          (engine.MODE_RAW, False)):
        y = torch.full((m.n_b, K), 3.0, dtype=torch.float64, device=dev)
        mask = torch.empty((m.n_b, K), dtype=torch.uint8, device=dev)
        engine.apply_strided(plan, xd, y, n_batch=1, k_inner=K,
                             x_row_stride=K, x_batch_stride=0,
                             y_row_stride=K, y_batch_stride=0, mode=emode,
                             threshold=0.1, mask_out=mask,
                             tune=[5, 512 if K == 64 else 0, 0, 0, 0, 0, 0, 0])
        if emode == engine.MODE_RAW:
            ref = oracle.csr_matvecs(csr, np.nan_to_num(x) * 0 + x)
            ref_mask = np.zeros_like(ref, dtype=bool)
        else:
            ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, masked,
                                              0.1)
            ref[ref_mask] = np.nan
        assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask)
        assert_bitwise(y.cpu().numpy(), ref, f'tile={tile} K={K}')


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('shape, axes', [
    ((1500, 128), [0]), ((1500, 64), [0]), ((1500, 130), [0]),
    ((2, 1500, 61), [1]), ((3, 1500, 64), [1]), ((2, 1500, 66), [1]),
    ((4, 1500, 33), [1]), ((1500, 63), [0]), ((2, 1500, 100), [1])])
def test_patch_kernel_f32_and_odd_strides(dev, shape, axes, dtype):
    """
    VERDICT round 2, item 6: the LDS patch kernel for f32 fields (the
    reference's real lat-lon input, tests/test_interpolate.py:492-516) and
    for odd strides / level counts -- (2, n, 61) -- which used to drop to
    `spmm_rowscalar`.  Forced through family 5 (`tune=[5]` without the hint
    flag fails loudly if the kernel cannot take the call), bitwise.
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.bilinear_map((30, 50), (70, 90), device=dev)
    assert m.n_a == 1500
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    choice = plan.auto_schedule(m.dst_dims)
    assert choice['family'] == 'patch', choice
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    rng = np.random.default_rng(shape[-1])
    x = rng.standard_normal(shape).astype(dtype)
    for masked in (False, True):
        if masked:
            dead = rng.random(m.n_a) < 0.2
            x[(slice(None),) * axes[0] + (dead,)] = np.nan
        arg = np.ma.masked_array(x, np.isnan(x)) if masked else x
        ref = oracle.remap_numpy_array(csr, frac_b, m.dst_dims, arg, axes,
                                       0.05 if masked else None).filled(
            np.nan)
        for view in ('aligned', 'offset'):
            xd = torch.from_numpy(x).to(dev)
            if view == 'offset':
                # a view that starts one element into its allocation
                big = torch.empty(xd.numel() + 1, dtype=xd.dtype, device=dev)
                big[1:] = xd.reshape(-1)
                xd = big[1:].reshape(shape)
            # (tune[1] = 512: the 512-thread workgroups long work lists of
            # 64-column chunks take; ignored by the 128-column chunks)
            for tune in ([5], [5, 512]):
                y = engine.remap_tensor(
                    plan, m.dst_dims, xd, axes,
                    engine.MODE_MASKED if masked else engine.MODE_FRACB,
                    threshold=0.05, tune=tune)
                assert_bitwise(y.cpu().numpy(), ref,
                               f'{shape} {dtype.__name__} {view} {masked} '
                               f'{tune}')


def test_patch_kernel_layouts_and_fallbacks(dev):
    """(T, n, L) in place through the patch family; 1-D destinations; the
    automatic choice falls back where the patch plan does not apply."""
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(900, (16, 24), 1, 6, seed=2)
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(mm['row'], mm['col'], mm['S'],
                                          mm['frac_b'], m.n_a, m.n_b,
                                          device=dev)
    plan.build_patches(m.dst_dims, tile=(4, 8))
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    rng = np.random.default_rng(0)
    f = rng.standard_normal((3, m.n_a, 66))
    ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, f, [1],
                                   None).filled(np.nan)
    y = engine.remap_tensor(plan, m.dst_dims, torch.from_numpy(f).to(dev),
                            [1], engine.MODE_FRACB)
    assert_bitwise(y.cpu().numpy(), ref)
    # float32 input: LDS patches too (256 / 512-byte staged rows)
    f32 = f.astype(np.float32)
    ref32 = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, f32, [1],
                                     None).filled(np.nan)
    y = engine.remap_tensor(plan, m.dst_dims, torch.from_numpy(f32).to(dev),
                            [1], engine.MODE_FRACB)
    assert_bitwise(y.cpu().numpy(), ref32)
    # 1-D destination: consecutive rows form the patches
    plan.build_patches(None, tile=(1, 32))
    assert plan.row_order is None and plan.patches['rows'] <= 32
    x = rng.standard_normal((m.n_a, 256))
    ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, False, 0.0)
    ref[ref_mask] = np.nan
    y = engine.remap_tensor(plan, (m.n_b,), torch.from_numpy(x).to(dev), [0],
                            engine.MODE_FRACB, tune=[5])
    assert_bitwise(y.cpu().numpy(), ref)
    # a budget nothing fits -> no patches, everything still works
    assert plan.build_patches(m.dst_dims, tile=(4, 8), lds_budget=512) is None
    y = engine.remap_tensor(plan, (m.n_b,), torch.from_numpy(x).to(dev), [0],
                            engine.MODE_FRACB)
    assert_bitwise(y.cpu().numpy(), ref)


# ---------------------------------------------------------------------------
# BASELINE.json's metric configuration at full size
# ---------------------------------------------------------------------------

def test_config3_full_size_bitwise_and_properties(dev):
    """
    EC30to60 -> 0.5 deg, 512 fp64 fields (2.04 GB per launch): bit-for-bit
    against the oracle, plus size-independent properties of the operator --
    linearity in the field, a constant field maps to the constant wherever
    rows are non-empty (rows sum to frac_b), masked rows are exactly the
    frac_b = 0 rows.
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.make_config('config3', device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    K = 512
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randn((m.n_a, K), generator=g, device=dev,
                    dtype=torch.float64)
    y = engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB)
    assert tuple(y.shape) == (360, 720, K)
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    ref, ref_mask = oracle.remap_flat(csr, frac_b, x.cpu().numpy(), False,
                                      0.0, nthreads=os.cpu_count() or 1)
    ref[ref_mask] = np.nan
    assert_bitwise(y.cpu().numpy().reshape(m.n_b, K), ref, 'config3 full')
    del ref
    # masked rows <=> frac_b == 0
    y2 = y.reshape(m.n_b, K)
    assert torch.equal(torch.isnan(y2[:, 0]), m.frac_b <= 0)
    # linearity: A(2x + 3z) == 2 A x + 3 A z up to rounding
    z = torch.randn((m.n_a, K), generator=g, device=dev,
                    dtype=torch.float64)
    yz = engine.remap_tensor(plan, m.dst_dims, z, [0], engine.MODE_FRACB)
    ylin = engine.remap_tensor(plan, m.dst_dims, 2 * x + 3 * z, [0],
                               engine.MODE_FRACB)
    ok = ~torch.isnan(ylin)
    err = (ylin - (2 * y + 3 * yz))[ok].abs().max()
    assert float(err) < 1e-12
    # a constant field is reproduced where the row is covered
    ones = torch.full((m.n_a, 128), 7.25, device=dev, dtype=torch.float64)
    yc = engine.remap_tensor(plan, m.dst_dims, ones, [0], engine.MODE_FRACB)
    okc = ~torch.isnan(yc)
    assert float((yc[okc] - 7.25).abs().max()) < 1e-12
    # every kernel family agrees bit for bit at this size
    plan.build_patches(m.dst_dims, tile=(4, 8))
    for tune in ([1, 2, 2, 4, 2], [1, 2, 1, 4, 1], [6, 2, 2, 4, 2], [5]):
        yt = engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB,
                                 tune=tune)
        assert torch.equal(torch.nan_to_num(yt, nan=1e300),
                           torch.nan_to_num(y, nan=1e300)), tune


@pytest.mark.parametrize('name, mode, locality', [
    ('config3', 'masked', 'raster'), ('headline', 'fracb', 'raster'),
    ('config4', 'fracb', 'raster'), ('config5', 'fracb', 'raster'),
    ('config5', 'masked', 'raster'), ('config3', 'masked', 'mesh'),
    ('headline', 'fracb', 'mesh'), ('config5', 'fracb', 'mesh'),
    ('config5', 'masked', 'scatter')])
def test_full_size_sampled_rows_bitwise(dev, name, mode, locality):
    """
    Every BASELINE configuration at its FULL size, as `Remapper` would run it
    (auto-selected schedule): a few thousand destination rows -- the first,
    the last and a random sample -- are recomputed by the CPU oracle from
    their own CSR rows and must match bit for bit; and the plain
    wave-per-row kernel must agree with the scheduled one on EVERY row.
    ``locality``: how the synthetic source mesh is numbered -- along the
    destination raster, as an MPAS mesh numbers its cells
    (``synthetic.mesh_numbering``), or at random.
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.make_config(name, device=dev, locality=locality)
    K = synthetic.CONFIGS[name]['K']
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    choice = plan.auto_schedule(m.dst_dims)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    x = torch.randn((m.n_a, K), generator=g, device=dev,
                    dtype=torch.float64)
    masked = mode == 'masked'
    if masked:
        x[torch.rand(m.n_a, generator=g, device=dev) < 0.2, :] = float('nan')
        x[torch.rand(m.n_a, generator=g, device=dev) < 0.05, ::7] = \
            float('nan')
    emode = engine.MODE_MASKED if masked else engine.MODE_FRACB
    y = engine.remap_tensor(plan, m.dst_dims, x, [0], emode, threshold=0.01)
    y = y.reshape(m.n_b, K)
    # -- sampled rows against the oracle ----------------------------------
    edge = torch.arange(64, device=dev)
    rows = torch.cat([edge, m.n_b - 1 - edge,
                      torch.randint(0, m.n_b, (4000,), generator=g,
                                    device=dev)]).unique()
    starts = plan.rowptr[rows]
    lens = plan.rowptr[rows + 1] - starts
    total = int(lens.sum())
    first = torch.cumsum(lens, 0) - lens
    idx = torch.repeat_interleave(starts - first, lens) + \
        torch.arange(total, device=dev)
    ucols, inv = torch.unique(plan.col[idx].to(torch.int64),
                              return_inverse=True)
    indptr = np.zeros(len(rows) + 1, dtype=np.int64)
    indptr[1:] = torch.cumsum(lens, 0).cpu().numpy()
    sub = oracle.OracleCSR(indptr, inv.cpu().numpy().astype(np.int32),
                           plan.val[idx].cpu().numpy(),
                           (len(rows), len(ucols)))
    ref, ref_mask = oracle.remap_flat(
        sub, m.frac_b[rows].cpu().numpy(), x[ucols].cpu().numpy(), masked,
        0.01, nthreads=8)
    ref[ref_mask] = np.nan
    assert_bitwise(y[rows].cpu().numpy(), ref,
                   f'{name} {mode} {choice["family"]}')
    assert bool(torch.isnan(y[rows]).any()) == bool(ref_mask.any())
    del ref
    # -- the unscheduled kernel agrees everywhere ---------------------------
    y1 = engine.remap_tensor(plan, m.dst_dims, x, [0], emode, threshold=0.01,
                             tune=[1]).reshape(m.n_b, K)
    same = (y1 == y) | (torch.isnan(y1) & torch.isnan(y))
    assert bool(same.all()), f'{name} {mode}: families disagree'


def test_apply_is_graph_capturable(dev):
    """remap_apply_f64 allocates nothing and never synchronises: it can be
    captured in a HIP graph and replayed on new field contents."""
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(3000, (30, 40), 1, 6, seed=8, device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    K = 256
    x = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
    y = torch.empty((m.n_b, K), device=dev, dtype=torch.float64)

    def launch():
        engine.apply_strided(plan, x, y, n_batch=1, k_inner=K,
                             x_row_stride=K, x_batch_stride=0,
                             y_row_stride=K, y_batch_stride=0,
                             mode=engine.MODE_FRACB)
    launch()
    torch.cuda.synchronize()
    expect1 = y.clone()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        launch()                               # warm-up on the side stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            launch()
    torch.cuda.synchronize()
    x.mul_(-2.0)                               # new contents, same buffers
    y.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(y, nan=1e300),
                       torch.nan_to_num(-2.0 * expect1, nan=1e300))


def _random_map(n_a, dst_dims, per_row, seed, dev):
    """No structure at all: every destination row draws random source rows."""
    from types import SimpleNamespace
    rng = np.random.default_rng(seed)
    n_b = int(np.prod(dst_dims))
    row = np.repeat(np.arange(1, n_b + 1, dtype=np.int32), per_row)
    col = rng.integers(1, n_a + 1, size=row.size).astype(np.int32)
    S = rng.random(row.size)
    frac_b = 0.5 + 0.5 * rng.random(n_b)
    t = lambda a: torch.from_numpy(a).to(dev)
    return SimpleNamespace(row=t(row), col=t(col), S=t(S), frac_b=t(frac_b),
                           n_a=n_a, n_b=n_b, dst_dims=dst_dims)


def test_auto_schedule_picks_by_reuse(dev):
    """Patches for a coarse -> fine bilinear map (heavy source-row reuse), row
    groups where neighbouring rows share some source rows (in 32 x 32
    supertiles for entry-rich rows), the plain register-gather kernel when
    they share nothing.  Whatever is chosen is bit-exact in every mode, for
    widths the preferred kernel cannot serve (the tune is a hint), and on a
    row shard that schedules itself."""
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    cases = [
        (synthetic.bilinear_map((20, 30), (200, 300), seed=1, device=dev),
         'patch'),
        (synthetic.conservative_map(30000, (150, 200), 3, 7, seed=2,
                                    device=dev), 'rowgroup'),
        (synthetic.conservative_map(30000, (96, 160), 12, 24, seed=3,
                                    device=dev), 'rowgroup'),
        (_random_map(5000, (64, 96), 4, 4, dev), 'rowscalar'),
    ]
    rng = np.random.default_rng(0)
    for m, expect in cases:
        plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                              m.n_a, m.n_b, device=dev)
        choice = plan.auto_schedule(m.dst_dims)
        assert choice['family'] == expect, choice
        assert (plan.patches is not None) == (expect == 'patch')
        assert (plan.groups is not None) == (expect == 'rowgroup')
        if expect == 'rowgroup':
            # entry-rich rows (2nd-order stencils): the shared form -- one
            # union per 4 x 8 tile through LDS -- in the frac_b / raw modes
            rich = choice['rows_per_group'] == 8
            assert ('share' in plan.groups) == rich
            assert (choice.get('shared_by') == 4) == rich
            assert (plan.default_tune[engine.MODE_FRACB][5:] == [32]) == rich
            assert plan.default_tune[engine.MODE_MASKED][5:] in ([], [0])
        rowptr, col, val = plan.to_host_csr()
        csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
        frac_b = m.frac_b.cpu().numpy()
        for K, masked in ((192, False), (192, True), (24, False),
                          (129, True), (320, False)):
            x = rng.standard_normal((m.n_a, K))
            if masked:
                x[rng.random(x.shape) < 0.2] = np.nan
            ref, mask = oracle.remap_flat(csr, frac_b, x, masked, 0.05,
                                          nthreads=4)
            ref[mask] = np.nan
            xd = torch.from_numpy(x).to(dev)
            y = engine.remap_tensor(
                plan, m.dst_dims, xd, [0],
                engine.MODE_MASKED if masked else engine.MODE_FRACB,
                threshold=0.05)
            assert_bitwise(y.cpu().numpy().reshape(m.n_b, K), ref,
                           f'{expect} K={K} masked={masked}')
        # a shard schedules itself over its own rows of the global grid
        shard = plan.shard(1, 3)
        shard.auto_schedule(m.dst_dims)
        ys = engine.remap_tensor(shard, None, xd, [0], engine.MODE_FRACB)
        r0 = shard.row_offset
        assert_bitwise(ys.cpu().numpy(), ref[r0:r0 + shard.n_b],
                       f'{expect} shard')


# ---------------------------------------------------------------------------
# row-group family: 8 or 4 rows per wave over the union of their columns
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('rows', [8, 4])
@pytest.mark.parametrize('grid', ['2d', '1d'])
@pytest.mark.parametrize('K', [61, 64, 128, 130, 131, 320, 512])
def test_rowgroup_kernel_bitwise(dev, grid, K, rows):
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    # 1502 destination rows: the last group is partial; wide stencils
    m = synthetic.conservative_map(1500, (23, 40) if grid == '2d' else
                                   (1, 1502 // 1 if False else 920), 3, 14,
                                   seed=17, signed=True)
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(mm['row'], mm['col'], mm['S'],
                                          mm['frac_b'], m.n_a, m.n_b,
                                          device=dev)
    ratio = plan.build_groups(m.dst_dims if grid == '2d' else None,
                              rows=rows)
    assert 0 < ratio <= 1
    assert (plan.row_order is not None) == (grid == '2d')
    assert plan.groups['rows'] == rows
    # the weights are stored once each: nnz of them (+ the readable pad)
    assert plan.groups['w'].numel() == plan.nnz + 128
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    rng = np.random.default_rng(K)
    x = rng.standard_normal((m.n_a, K))
    x[rng.random(m.n_a) < 0.2, :] = np.nan
    xd = torch.from_numpy(x).to(dev)
    for emode, masked in ((engine.MODE_FRACB, False),
                          (engine.MODE_MASKED, True),
                          (engine.MODE_RAW, False)):
        if emode == engine.MODE_RAW:
            ref = oracle.csr_matvecs(csr, x)
            ref_mask = np.zeros_like(ref, dtype=bool)
        else:
            ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, masked,
                                              0.1)
            ref[ref_mask] = np.nan
        for tune in ([10], [10, 0, 2, 1], [10, 0, 1, 3, 1],
                     [10, 0, 1, 2, 0, 4], [10, 0, 2, 1, 0, 4],
                     [10, 1, 2, 1], [10, 2, 1, 3], [10, 1, 1, 2, 1],
                     [10, 0, 1, 1, 0, 16], [10, 1, 2, 2, 0, 16]):
            y = torch.full((m.n_b, K), 3.0, dtype=torch.float64, device=dev)
            mask = torch.full((m.n_b, K), 7, dtype=torch.uint8, device=dev)
            engine.apply_strided(plan, xd, y, n_batch=1, k_inner=K,
                                 x_row_stride=K, x_batch_stride=0,
                                 y_row_stride=K, y_batch_stride=0,
                                 mode=emode, threshold=0.1, mask_out=mask,
                                 tune=tune)
            assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask)
            assert_bitwise(y.cpu().numpy(), ref, f'{grid} K={K} {tune}')


# ---------------------------------------------------------------------------
# randomised end-to-end cases: whatever the plan, the layout and the mode
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('seed', range(12))
def test_random_cases_bitwise(dev, seed):
    """
    Twelve seeds x ten random cases: random mapping (bilinear-like,
    conservative-like or structureless), random destination grid shape,
    schedule chosen by ``auto_schedule`` (or none), random field layout (the
    source axis anywhere in a 1- to 4-D array), f32 or f64, the three modes,
    whole plan or a row shard -- always bit-identical to the oracle's
    restatement of ``_remap_numpy_array``.
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    rng = np.random.default_rng(1000 + seed)
    for case in range(10):
        my, mx = int(rng.integers(3, 40)), int(rng.integers(4, 60))
        kind = rng.choice(['bilinear', 'conservative', 'rich', 'random'])
        if kind == 'bilinear':
            sy, sx = int(rng.integers(2, 12)), int(rng.integers(2, 14))
            m = synthetic.bilinear_map((sy, sx), (my, mx),
                                       seed=int(rng.integers(1 << 30)),
                                       device=dev)
        elif kind == 'random':
            m = _random_map(int(rng.integers(5, 3000)), (my, mx),
                            int(rng.integers(1, 6)),
                            int(rng.integers(1 << 30)), dev)
        else:
            lo, hi = (1, 6) if kind == 'conservative' else (10, 22)
            m = synthetic.conservative_map(
                int(rng.integers(50, 4000)), (my, mx), lo, hi,
                seed=int(rng.integers(1 << 30)), device=dev,
                signed=bool(rng.integers(2)))
        plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                              m.n_a, m.n_b, device=dev)
        how = rng.choice(['auto', 'auto1d', 'none'])
        if how == 'auto':
            plan.auto_schedule((my, mx))
        elif how == 'auto1d':
            plan.auto_schedule((my * mx,))
        rowptr, col, val = plan.to_host_csr()
        csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
        frac_b = m.frac_b.cpu().numpy()
        # field layout: extra dims around the source axis
        n_before = int(rng.integers(0, 3))
        n_after = int(rng.integers(0, 3))
        before = [int(rng.integers(1, 5)) for _ in range(n_before)]
        after = [int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65, 130, 256]))
                 for _ in range(n_after)]
        shape = before + [m.n_a] + after
        axis = n_before
        dtype = rng.choice([np.float64, np.float32])
        field = rng.standard_normal(shape).astype(dtype)
        mode = rng.choice(['fracb', 'masked', 'plain_thr'])
        thr = None
        arg = field
        if mode != 'fracb':
            thr = float(rng.choice([0.0, 0.01, 0.5]))
        if mode == 'masked':
            field[rng.random(shape) < 0.15] = np.nan
            # `_remap_data_array` wraps a field as a MaskedArray only if it
            # holds a NaN (remap_numpy.py:201-204); a small field may have
            # drawn none (found by running 400 seeds: seed 188)
            if np.isnan(field).any():
                arg = np.ma.masked_array(field, mask=np.isnan(field))
        ref = oracle.remap_numpy_array(csr, frac_b, (my, mx), arg, [axis],
                                       thr)
        ref = np.ma.filled(ref.astype(np.float64), np.nan) \
            if np.ma.isMaskedArray(ref) else np.asarray(ref)
        x = torch.from_numpy(field).to(dev)
        masked = mode == 'masked' and thr is not None and \
            bool(np.isnan(field).any())
        y = engine.remap_tensor(
            plan, (my, mx), x, [axis],
            engine.MODE_MASKED if masked else engine.MODE_FRACB,
            threshold=thr if masked else 0.0)
        what = (f'seed {seed} case {case}: {kind} {m.n_a}->{my}x{mx} {how} '
                f'shape {shape} {np.dtype(dtype).name} {mode} thr {thr}')
        assert tuple(y.shape) == ref.shape, what
        assert_bitwise(y.cpu().numpy(), ref, what)
        if m.n_b >= 6:
            # a row shard of the same plan, scheduling itself
            r = int(rng.integers(0, 3))
            shard = plan.shard(r, 3)
            if how != 'none':
                shard.auto_schedule((my, mx))
            ys = engine.remap_tensor(
                shard, None, x, [axis],
                engine.MODE_MASKED if masked else engine.MODE_FRACB,
                threshold=thr if masked else 0.0)
            flat = ref.reshape(tuple(before) + (m.n_b,) + tuple(after))
            lo = shard.row_offset
            want = np.take(flat, np.arange(lo, lo + shard.n_b), axis=axis)
            assert_bitwise(ys.cpu().numpy(), want, what + f' shard {r}/3')


def test_sharded_remap_slabs_make_the_whole(dev):
    """``parallel.ShardedRemap`` as three ranks would build it (ranks given
    explicitly): every rank's slab, scheduled on its own, stacked = the
    unsharded result, bit for bit; the work-balanced bounds cover all rows."""
    from pyremap_amd import engine, synthetic
    from pyremap_amd.parallel import ShardedRemap
    m = synthetic.conservative_map(20000, (120, 150), 2, 7, seed=13,
                                   device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    x = torch.randn((4, m.n_a, 96), device=dev, dtype=torch.float64)
    x[:, torch.rand(m.n_a, device=dev) < 0.1, 40:] = float('nan')
    whole = ShardedRemap(plan, grid_dims=m.dst_dims)        # no group: 1 rank
    assert whole.world_size == 1 and whole.plan.n_b == plan.n_b
    want = engine.remap_tensor(plan, m.dst_dims, x, [1], engine.MODE_MASKED,
                               threshold=0.01)
    alone = whole.apply(whole.distribute(x, axis=1), [1], engine.MODE_MASKED,
                        threshold=0.01)
    assert_bitwise(alone.cpu().numpy(),
                   want.reshape(4, m.n_b, 96).cpu().numpy(), 'one rank')
    slabs, covered = [], 0
    for rank in range(3):
        part = ShardedRemap(plan, grid_dims=m.dst_dims, rank=rank,
                            world_size=3)
        assert part.bounds[0] == 0 and part.bounds[-1] == m.n_b
        assert part.plan.n_b == part.bounds[rank + 1] - part.bounds[rank]
        assert part.schedule is not None
        covered += part.plan.n_b
        # the shard lives in the compact space of the source rows it reads
        assert part.plan.n_a == part.ucols.shape[0] < m.n_a
        slabs.append(part.apply(engine.gather_rows(x, 1, part.ucols), [1],
                                engine.MODE_MASKED, threshold=0.01))
    assert covered == m.n_b
    got = torch.cat(slabs, dim=1)
    assert_bitwise(got.cpu().numpy(),
                   want.reshape(4, m.n_b, 96).cpu().numpy(), 'sharded')
    # work balance: no rank carries more than 40 % of entries + 2 rows each
    work = [int(plan.rowptr[b1] - plan.rowptr[b0]) + 2 * (b1 - b0)
            for b0, b1 in zip(part.bounds[:-1], part.bounds[1:])]
    assert max(work) < 0.4 * sum(work)


def test_engine_odds_and_ends(dev, problem):
    """Entry points the other tests do not reach: a plan from CSR arrays, a
    hand-made row order, the Morton walk, the byte accounting of SURVEY 8(d),
    the stream copy, ``out=`` on the permuting path, 0-based triplets, and
    the argument errors of the Python layer."""
    from oracle import oracle
    from pyremap_amd import Remapper, engine, synthetic
    from pyremap_amd.descriptor import MpasCellMeshDescriptor
    m = synthetic.conservative_map(3000, (40, 50), 1, 6, seed=31, device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    rowptr, col, val = plan.to_host_csr()
    frac_b = m.frac_b.cpu().numpy()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    x = torch.randn((m.n_a, 160), device=dev, dtype=torch.float64)
    ref, mask = oracle.remap_flat(csr, frac_b, x.cpu().numpy(), False, 0.0)
    ref[mask] = np.nan
    # from_csr == from_triplets
    again = engine.RemapPlan.from_csr(rowptr, col, val, frac_b, m.n_a,
                                      device=dev)
    assert (again.n_a, again.n_b, again.nnz) == (plan.n_a, plan.n_b,
                                                 plan.nnz)
    y = engine.remap_tensor(again, None, x, [0], engine.MODE_FRACB)
    assert_bitwise(y.cpu().numpy(), ref, 'from_csr')
    # processing orders never change results
    plan.set_row_order(np.arange(m.n_b)[::-1].copy())
    y = engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB)
    assert_bitwise(y.cpu().numpy().reshape(m.n_b, -1), ref, 'reversed')
    plan.set_grid_schedule(m.dst_dims, 'morton')
    y = engine.remap_tensor(plan, m.dst_dims, x, [0], engine.MODE_FRACB)
    assert_bitwise(y.cpu().numpy().reshape(m.n_b, -1), ref, 'morton')
    plan.set_row_order(None)
    assert plan.row_order is None
    with pytest.raises(ValueError, match='row order must have'):
        plan.set_row_order(np.arange(5))
    with pytest.raises(ValueError, match='unknown schedule'):
        plan.set_grid_schedule(m.dst_dims, 'hilbert')
    with pytest.raises(ValueError, match='does not hold'):
        plan.set_grid_schedule((7, 9), 'tile', (2, 2))
    with pytest.raises(ValueError, match='do not hold n_b'):
        engine.remap_tensor(plan, (7, 9), x, [0], engine.MODE_FRACB)
    # S + col, rowptr, X once, Y once, frac_b
    K = 160
    assert plan.algorithmic_bytes(K) == plan.nnz * 12 + (m.n_b + 1) * 8 + \
        m.n_a * K * 8 + m.n_b * K * 8 + m.n_b * 8
    assert plan.algorithmic_bytes(K, 4, engine.MODE_MASKED) == \
        plan.nnz * 12 + (m.n_b + 1) * 8 + m.n_a * K * 4 + m.n_b * K * 8
    # the ceiling probe copies
    src = torch.randint(0, 255, (1 << 20,), dtype=torch.uint8, device=dev)
    dst = torch.zeros_like(src)
    engine.stream_copy(dst, src)
    assert torch.equal(dst, src)
    # out= when the source axis is not leading enough to stride (permute path)
    f = torch.randn((m.n_a, 3, 5), device=dev, dtype=torch.float64)
    want = engine.remap_tensor(plan, m.dst_dims, f, [0], engine.MODE_FRACB)
    g = f.permute(1, 0, 2).contiguous()              # (3, n_a, 5): k_inner 5
    out = torch.empty((3,) + tuple(m.dst_dims) + (5,), device=dev,
                      dtype=torch.float64)
    got = engine.remap_tensor(plan, m.dst_dims, g, [1], engine.MODE_FRACB,
                              out=out)
    assert got.data_ptr() == out.data_ptr()
    assert_bitwise(got.permute(1, 2, 0, 3).cpu().numpy(),
                   want.cpu().numpy(), 'out= on the permute path')
    # argument errors
    with pytest.raises(ValueError, match='same length'):
        engine.RemapPlan.from_triplets(m.row[:-1], m.col, m.S, m.frac_b,
                                       m.n_a, m.n_b, device=dev)
    with pytest.raises(ValueError, match='frac_b has'):
        engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b[:-1],
                                       m.n_a, m.n_b, device=dev)
    # other dtypes are upcast first, as scipy does; the raw entry refuses
    xh = x.to(torch.float16)
    yh = engine.remap_tensor(plan, m.dst_dims, xh, [0], engine.MODE_FRACB)
    yd = engine.remap_tensor(plan, m.dst_dims, xh.to(torch.float64), [0],
                             engine.MODE_FRACB)
    assert_bitwise(yh.cpu().numpy(), yd.cpu().numpy(), 'f16 upcast')
    with pytest.raises(TypeError, match='float64 or float32'):
        engine.apply_strided(plan, xh, yd.reshape(m.n_b, K), n_batch=1,
                             k_inner=K, x_row_stride=K, x_batch_stride=0,
                             y_row_stride=K, y_batch_stride=0,
                             mode=engine.MODE_FRACB)
    # 0-based triplets through the Remapper front door
    host = m.numpy()
    src_d = MpasCellMeshDescriptor(mesh_name='toy', size=m.n_a)
    from pyremap_amd import LatLonGridDescriptor
    dst_d = LatLonGridDescriptor.create(np.linspace(-90, 90, 41),
                                        np.linspace(-180, 180, 51))
    r0 = Remapper.from_triplets(host['row'] - 1, host['col'] - 1, host['S'],
                                host['frac_b'], src_d, dst_d, index_base=0)
    y0 = r0.remap_array(x, [0])
    assert_bitwise(y0.cpu().numpy().reshape(m.n_b, -1), ref, 'index_base 0')
    with pytest.raises(ValueError, match='No mapping file'):
        Remapper(src_descriptor=src_d, dst_descriptor=dst_d).remap_array(
            x, [0])


def test_ncremap_reads_netcdf4_input(tmp_path):
    """A NetCDF-4 field file through ``ncremap``: read by the package's HDF5
    reader and answered in NetCDF-4 by its HDF5 writer (as NCO keeps the
    input's format), same numbers as ``remap_numpy``."""
    from pyremap_amd import (
        LatLonGridDescriptor,
        MpasCellMeshDescriptor,
        Remapper,
        synthetic,
    )
    from pyremap_amd.io.netcdf import file_format, open_dataset
    src_file = os.path.join(os.path.dirname(__file__), 'golden', 'hdf5',
                            'nc4_ref_latlon_to_mpas_cell.nc')
    assert file_format(src_file) == 'NETCDF4'
    n_cells = 7153
    mm = synthetic.conservative_map(n_cells, (12, 24), 1, 5, seed=17)
    map_path = str(tmp_path / 'map.nc')
    mm.save(map_path)
    r = Remapper(map_filename=map_path,
                 src_descriptor=MpasCellMeshDescriptor(mesh_name='qu240',
                                                       size=n_cells),
                 dst_descriptor=LatLonGridDescriptor.create(
                     np.linspace(-90, 90, 13), np.linspace(-180, 180, 25)))
    out_path = str(tmp_path / 'out.nc')
    r.ncremap(src_file, out_path, variable_list=['SST'])
    assert file_format(out_path) == 'NETCDF4'
    on_disk = open_dataset(out_path)
    in_mem = r.remap_numpy(open_dataset(src_file))
    assert on_disk['SST'].dims == ('time', 'lat', 'lon')
    assert_bitwise(on_disk['SST'].values, in_mem['SST'].values, 'nc4 in')
    with pytest.raises(ValueError, match='are not in'):
        r.ncremap(src_file, str(tmp_path / 'o2.nc'),
                  variable_list=['no_such_variable'])


@pytest.mark.parametrize('name, mode', [
    ('headline', 'fracb'), ('config4', 'fracb'), ('config5', 'fracb'),
    ('config5', 'masked')])
def test_full_size_every_value_bitwise(dev, name, mode):
    """
    The large BASELINE configurations at FULL size with EVERY value compared
    with the oracle (round 2 compared ~4 000 sampled rows and the rest only
    kernel against kernel): the fields go through the scheduled kernel once,
    at the configuration's own K, and the result is checked in column slices
    of 16 fields -- the oracle on all host cores per slice -- so that the
    host never holds more than one 16-field slab of the result (config 4:
    3.8 GB of its 30.7 GB).
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.make_config(name, device=dev, locality='mesh')
    K = synthetic.CONFIGS[name]['K']
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    del m.row, m.col, m.S
    plan.auto_schedule(m.dst_dims)
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    x = torch.randn((m.n_a, K), generator=g, device=dev,
                    dtype=torch.float64)
    masked = mode == 'masked'
    if masked:
        x[torch.rand(m.n_a, generator=g, device=dev) < 0.2, :] = float('nan')
        x[torch.rand(m.n_a, generator=g, device=dev) < 0.05, ::7] = \
            float('nan')
    y = engine.remap_tensor(
        plan, m.dst_dims, x, [0],
        engine.MODE_MASKED if masked else engine.MODE_FRACB, threshold=0.01)
    y = y.reshape(m.n_b, K)
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    threads = os.cpu_count() or 1
    step = 16
    for k0 in range(0, K, step):
        xs = x[:, k0:k0 + step].contiguous().cpu().numpy()
        ref, ref_mask = oracle.remap_flat(csr, frac_b, xs, masked, 0.01,
                                          nthreads=threads)
        got = y[:, k0:k0 + step].contiguous().cpu().numpy()
        # bit patterns wherever the oracle does not mask; NaN where it does
        assert np.array_equal(np.isnan(got), ref_mask | np.isnan(ref)), \
            f'{name} {mode}: NaN placement, fields {k0}..'
        ok = ~np.isnan(got)
        assert np.array_equal(got.view(np.int64)[ok],
                              ref.view(np.int64)[ok]), \
            f'{name} {mode}: values differ in fields {k0}..{k0 + step}'
        del xs, ref, ref_mask, got, ok


# ---------------------------------------------------------------------------
# BASELINE.json configs 1 and 2 at their own sizes
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('mode', ['fracb', 'masked'])
def test_config2_full_size_bitwise(dev, mode):
    """
    BASELINE config 2 -- QU240 (7 153 cells) -> 1 deg (180 x 360)
    conservative, 64 batched fp64 fields -- as `Remapper` runs it (the
    auto-selected schedule; on this map that is the LDS patch family at
    K = 64): EVERY row bit for bit against the oracle, in the unmasked and
    the NaN-masked + renormalised mode, as an (n_a, K) field and as the
    MPAS layout (Time, nCells, nVertLevels) addressed in place.
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.make_config('config2', device=dev)
    K = synthetic.CONFIGS['config2']['K']
    assert (m.n_a, m.n_b, K) == (7153, 180 * 360, 64)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    choice = plan.auto_schedule(m.dst_dims)
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    rng = np.random.default_rng(22)
    masked = mode == 'masked'
    emode = engine.MODE_MASKED if masked else engine.MODE_FRACB
    # (n_a, K)
    x = rng.standard_normal((m.n_a, K))
    if masked:
        x[rng.random(m.n_a) < 0.2, :] = np.nan
        x[rng.random(m.n_a) < 0.1, ::5] = np.nan
    arg = np.ma.masked_array(x, np.isnan(x)) if masked else x
    ref = oracle.remap_numpy_array(csr, frac_b, m.dst_dims, arg, [0],
                                   0.01 if masked else None)
    y = engine.remap_tensor(plan, m.dst_dims, torch.from_numpy(x).to(dev),
                            [0], emode, threshold=0.01)
    assert tuple(y.shape) == (180, 360, K)
    assert_bitwise(y.cpu().numpy(), np.ma.filled(ref, np.nan),
                   f'config2 {mode} {choice["family"]}')
    # (Time, nCells, nVertLevels) = (4, 7153, 16): 64 fields in place
    x3 = rng.standard_normal((4, m.n_a, 16))
    if masked:
        x3[:, rng.random(m.n_a) < 0.2, 9:] = np.nan
    arg = np.ma.masked_array(x3, np.isnan(x3)) if masked else x3
    ref = oracle.remap_numpy_array(csr, frac_b, m.dst_dims, arg, [1],
                                   0.01 if masked else None)
    y = engine.remap_tensor(plan, m.dst_dims, torch.from_numpy(x3).to(dev),
                            [1], emode, threshold=0.01)
    assert tuple(y.shape) == (4, 180, 360, 16)
    assert_bitwise(y.cpu().numpy(), np.ma.filled(ref, np.nan),
                   f'config2 {mode} layout (T, nCells, L)')
    # the unscheduled kernel agrees
    y1 = engine.remap_tensor(plan, m.dst_dims, torch.from_numpy(x3).to(dev),
                             [1], emode, threshold=0.01, tune=[1])
    assert torch.equal(torch.nan_to_num(y1, nan=1e300),
                       torch.nan_to_num(y, nan=1e300))


def test_config1_end_to_end(dev, tmp_path):
    """
    BASELINE config 1 -- 1 deg -> 0.5 deg lat-lon bilinear, ONE 2-D field --
    end to end as a pyremap user runs it
    (reference tests/test_interpolate.py:492-516, remap_numpy.py:236-256 with
    K = 1): descriptors from `get_lat_lon_descriptor`, the mapping file on
    disk under the reference's default name, `Remapper(map_filename=...)`,
    then `remap_numpy(DataArray)`, `remap_numpy(Dataset)` with a threshold
    and `ncremap()` file -> file.  Every value bit for bit against the oracle
    on the same triplets.
    """
    from oracle import oracle
    from pyremap_amd import (
        DataArray,
        Dataset,
        Remapper,
        get_lat_lon_descriptor,
        synthetic,
    )
    from pyremap_amd.io.netcdf import open_dataset, write_netcdf
    src = get_lat_lon_descriptor(dlon=1.0, dlat=1.0)
    dst = get_lat_lon_descriptor(dlon=0.5, dlat=0.5)
    assert src.dim_sizes == [180, 360] and dst.dim_sizes == [360, 720]
    m = synthetic.make_config('config1')
    assert (m.n_a, m.n_b) == (64800, 259200)
    map_path = str(tmp_path / f'map_{src.mesh_name}_to_{dst.mesh_name}_'
                              f'bilinear.nc')
    m.save(map_path)
    mm = m.numpy()
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    r = Remapper(map_filename=map_path, src_descriptor=src,
                 dst_descriptor=dst)
    rng = np.random.default_rng(101)
    lat = np.deg2rad(src.lat)[:, None]
    lon = np.deg2rad(src.lon)[None, :]
    field = np.cos(lat) * np.sin(2 * lon) + \
        0.01 * rng.standard_normal((180, 360))
    # -- one 2-D field, no NaN, no threshold ------------------------------
    da = DataArray(field, dims=('lat', 'lon'), name='sst',
                   attrs={'units': 'K'})
    out = r.remap_numpy(da)
    assert out.dims == ('lat', 'lon') and out.values.shape == (360, 720)
    assert out.values.dtype == np.float64
    # remap_numpy.py:60-69: history and mesh_name are added to the result
    assert out.attrs['units'] == 'K' and \
        out.attrs['mesh_name'] == '0.5x0.5degree' and 'history' in out.attrs
    ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, field,
                                   [0, 1], None)
    assert_bitwise(out.values, np.ma.filled(ref, np.nan), 'config1 2-D')
    np.testing.assert_array_equal(out.coords['lat'].values, dst.lat)
    # a bilinear map of a smooth field stays close to it
    assert np.abs(out.values).max() <= np.abs(field).max() + 1e-12
    # -- Dataset: land as NaN + renormalisation, a float32 field, a
    #    (time, lat, lon) field and a pass-through variable ---------------
    landed = field.copy()
    landed[40:60, 100:180] = np.nan
    f32 = rng.standard_normal((180, 360)).astype(np.float32)
    monthly = rng.standard_normal((12, 180, 360))
    ds = Dataset(attrs={'title': 'config 1'})
    ds['sst'] = DataArray(landed, dims=('lat', 'lon'))
    ds['ice'] = DataArray(f32, dims=('lat', 'lon'))
    ds['monthly'] = DataArray(monthly, dims=('time', 'lat', 'lon'))
    ds['month'] = DataArray(np.arange(12), dims=('time',))
    out = r.remap_numpy(ds, renormalization_threshold=0.05)
    ref = oracle.remap_numpy_array(
        csr, mm['frac_b'], m.dst_dims,
        np.ma.masked_array(landed, np.isnan(landed)), [0, 1], 0.05)
    assert_bitwise(out['sst'].values, np.ma.filled(ref, np.nan),
                   'config1 masked')
    assert np.isnan(out['sst'].values).any()
    ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, f32,
                                   [0, 1], 0.05)
    assert_bitwise(out['ice'].values, np.ma.filled(ref, np.nan),
                   'config1 f32')
    ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, monthly,
                                   [1, 2], 0.05)
    assert out['monthly'].dims == ('time', 'lat', 'lon')
    assert_bitwise(out['monthly'].values, np.ma.filled(ref, np.nan),
                   'config1 (time, lat, lon)')
    np.testing.assert_array_equal(out['month'].values, np.arange(12))
    # -- file -> file -----------------------------------------------------
    in_path = str(tmp_path / 'in.nc')
    out_path = str(tmp_path / 'out.nc')
    write_netcdf(ds, in_path)
    r.ncremap(in_path, out_path, renormalize=0.05)
    back = open_dataset(out_path)
    for name in ('sst', 'ice', 'monthly', 'month'):
        assert_bitwise(back[name].values, out[name].values,
                       f'config1 file {name}')


# ---------------------------------------------------------------------------
# the few-fields path (K <= 32): one 2-D field, (Time, nCells) monthly data
# ---------------------------------------------------------------------------

FEW_LAYOUTS = [
    # shape builder (n_a -> shape), remap axis
    ('n', lambda n: (n,), 0),                    # one field, K = 1
    ('nk5', lambda n: (n, 5), 0),
    ('tn12', lambda n: (12, n), 1),              # (Time, nCells)
    ('tnl', lambda n: (3, n, 7), 1),             # (Time, nCells, 7 levels)
    ('nk32', lambda n: (n, 32), 0),
]


@pytest.mark.parametrize('tune', [None, [2], [2, 1], [2, 4], [3], [3, 4],
                                  [3, 8]])
@pytest.mark.parametrize('layout', FEW_LAYOUTS, ids=[f[0] for f in
                                                    FEW_LAYOUTS])
def test_few_fields_kernels_bitwise(problem, dev, layout, tune):
    """
    remap_numpy.py:240-256 with few batched fields: both small-K kernel
    families (lane per (row, k); sub-group of 8 or 4 lanes per row with the
    products added in CSR order), f32 and f64, three modes, every layout
    addressed IN PLACE -- bit for bit the reference's values.
    """
    from oracle import oracle
    from pyremap_amd import engine
    p = problem
    _, shape_of, axis = layout
    rng = np.random.default_rng(len(shape_of(3)) * 7 + axis)
    for dtype in (np.float64, np.float32):
        x = rng.standard_normal(shape_of(p['n_a'])).astype(dtype)
        dead = rng.random(p['n_a']) < 0.2
        xm = x.copy()
        xm[(slice(None),) * axis + (dead,)] = np.nan
        for emode, masked, field in ((engine.MODE_FRACB, False, x),
                                     (engine.MODE_MASKED, True, xm)):
            arg = np.ma.masked_array(field, np.isnan(field)) if masked \
                else field
            ref = oracle.remap_numpy_array(
                p['csr'], p['frac_b'], (p['n_b'],), arg, [axis],
                0.1 if masked else None)
            y = engine.remap_tensor(p['plan'], (p['n_b'],),
                                    torch.from_numpy(field).to(dev), [axis],
                                    emode, threshold=0.1, tune=tune)
            assert_bitwise(y.cpu().numpy(), np.ma.filled(ref, np.nan),
                           f'{layout[0]} {dtype.__name__} {emode} {tune}')


def test_few_fields_are_addressed_in_place(problem, dev, monkeypatch):
    """A (Time, nCells) field takes ONE launch and no permute copy."""
    from pyremap_amd import engine
    p = problem
    calls = []
    real = engine.apply_strided

    def spy(plan, X, Y, **kw):
        calls.append((X.data_ptr(), kw['n_batch'], kw['k_inner'],
                      kw['x_row_stride'], kw['x_batch_stride']))
        return real(plan, X, Y, **kw)
    monkeypatch.setattr(engine, 'apply_strided', spy)
    x = torch.randn((12, p['n_a']), dtype=torch.float64, device=dev)
    engine.remap_tensor(p['plan'], (p['n_b'],), x, [1], engine.MODE_FRACB)
    assert calls == [(x.data_ptr(), 12, 1, 1, p['n_a'])]


def test_tree_flag_is_close_not_identical(problem, dev):
    """REMAP_FLAG_TREE (opt-in): butterfly sums, rtol 1e-13, not the bits."""
    from oracle import oracle
    from pyremap_amd import engine
    p = problem
    rng = np.random.default_rng(5)
    x = rng.standard_normal((p['n_a'], 3))
    ref, ref_mask = oracle.remap_flat(p['csr'], p['frac_b'], x, False, 0.0)
    xd = torch.from_numpy(x).to(dev)
    ys = {}
    for flags in (0, engine.FLAG_TREE):
        y = engine.remap_tensor(p['plan'], (p['n_b'],), xd, [0],
                                engine.MODE_FRACB, flags=flags, tune=[3])
        ys[flags] = y.cpu().numpy()
    ok = ~ref_mask
    assert_bitwise(np.where(ok, ys[0], 0.0), np.where(ok, ref, 0.0))
    scale = np.abs(ref[ok]).max()
    assert np.abs(ys[engine.FLAG_TREE][ok] - ref[ok]).max() <= 1e-13 * scale
    assert np.array_equal(np.isnan(ys[engine.FLAG_TREE]), ref_mask)
    # long rows make the association visible
    assert not np.array_equal(ys[engine.FLAG_TREE][ok], ref[ok])


# ---------------------------------------------------------------------------
# device-side branch selection (remap_numpy.py:201-204) and the host path
# ---------------------------------------------------------------------------

def test_scan_nan_and_gated_launches(problem, dev):
    """remap_scan_nan + remap_apply_args.gate: `isnan(values).any()` decides
    the branch on the device; the launch whose gate is closed writes
    nothing."""
    from oracle import oracle
    from pyremap_amd import engine
    p = problem
    for dtype in (torch.float64, torch.float32):
        for n in (1, 3, 64, 1027, 70001):
            x = torch.randn(n, dtype=dtype, device=dev)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            engine.scan_nan(x, flag)
            assert int(flag) == 0
            x[n - 1] = float('nan')       # the tail element
            engine.scan_nan(x, flag)
            assert int(flag) == 1
            x[n - 1] = 0.0
            x[n // 2] = float('nan')
            flag.zero_()
            engine.scan_nan(x, flag)
            assert int(flag) == 1
    rng = np.random.default_rng(77)
    for K in (3, 200):
        clean = rng.standard_normal((p['n_a'], K))
        holed = clean.copy()
        holed[rng.random(p['n_a']) < 0.3, :] = np.nan
        for x, masked in ((clean, False), (holed, True)):
            xd = torch.from_numpy(x).to(dev)
            y = engine.remap_tensor_auto_mode(p['plan'], (p['n_b'],), xd,
                                              [0], 0.2)
            ref, ref_mask = oracle.remap_flat(p['csr'], p['frac_b'], x,
                                              masked, 0.2)
            ref[ref_mask] = np.nan
            assert_bitwise(y.cpu().numpy(), ref, f'auto mode K={K} {masked}')
        # a closed gate leaves the output untouched
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        y = torch.full((p['n_b'], K), 5.0, dtype=torch.float64, device=dev)
        engine.remap_tensor(p['plan'], (p['n_b'],), xd, [0],
                            engine.MODE_FRACB, out=y, gate=flag,
                            gate_value=1)
        assert bool((y == 5.0).all())


def test_host_arrays_pinned_pipelined_and_poisoned(dev):
    """
    numpy in -> numpy out through pyremap_amd.host_path: the single-shot and
    the three-stream pipelined form (leading batch dims), NaN-decided branch,
    masks, and a MaskedArray whose data holds an UNMASKED NaN (the reference
    lets it through, remap_numpy.py:263) -- all against the oracle.
    """
    from oracle import oracle
    from pyremap_amd import engine, host_path, synthetic
    m = synthetic.conservative_map(3000, (40, 50), 1, 6, seed=3)
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(mm['row'], mm['col'], mm['S'],
                                          mm['frac_b'], m.n_a, m.n_b,
                                          device=dev)
    plan.auto_schedule(m.dst_dims)
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((6, m.n_a, 40))
    holed = x.copy()
    holed[:, rng.random(m.n_a) < 0.2, 10:] = np.nan
    old = host_path.CHUNK_BYTES
    host_path.CHUNK_BYTES = 2 * m.n_a * 40 * 8      # three chunks of two
    try:
        for field, mode, thr in ((x, 'fracb', None), (holed, 'masked', 0.3),
                                 (holed, 'auto', 0.3), (x, 'auto', 0.3),
                                 (x.astype(np.float32), 'fracb', None)):
            masked = mode == 'masked' or (mode == 'auto' and
                                          np.isnan(field).any())
            arg = np.ma.masked_array(field, np.isnan(field)) if masked \
                else field
            ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims,
                                           arg, [1], thr if masked else None)
            want_mask = mode != 'auto'
            got = host_path.remap_host_array(
                plan, m.dst_dims, field, [1], mode=mode, threshold=thr,
                want_mask=want_mask).result()
            data = got[0] if want_mask else got
            assert_bitwise(data, np.ma.filled(ref, np.nan),
                           f'host path {mode} {field.dtype}')
            if want_mask:
                assert np.array_equal(got[1], np.ma.getmaskarray(ref))
    finally:
        host_path.CHUNK_BYTES = old
    # more chunks than ring slots: the device buffers are reused while the
    # series streams through (7 batches, one per chunk, two slots)
    old_ring = host_path.RING_SLOTS
    host_path.CHUNK_BYTES = m.n_a * 40 * 8
    host_path.RING_SLOTS = 2
    try:
        series = rng.standard_normal((7, m.n_a, 40))
        series[3:, rng.random(m.n_a) < 0.3, 20:] = np.nan
        ref = oracle.remap_numpy_array(
            csr, mm['frac_b'], m.dst_dims,
            np.ma.masked_array(series, np.isnan(series)), [1], 0.2)
        data, mask = host_path.remap_host_array(
            plan, m.dst_dims, series, [1], mode='masked', threshold=0.2,
            want_mask=True).result()
        assert_bitwise(data, np.ma.filled(ref, np.nan), 'ring of two')
        assert np.array_equal(mask, np.ma.getmaskarray(ref))
        # 'auto' on a field whose NaNs a strided sample finds: the branch is
        # known before the upload, the field streams (no device-side wait)
        seen = []
        real_enqueue = host_path._enqueue

        def spy(plan_, dims_, values_, host_, axes_, lead_, nb_, inpl_,
                mode_, *rest):
            seen.append(mode_)
            return real_enqueue(plan_, dims_, values_, host_, axes_, lead_,
                                nb_, inpl_, mode_, *rest)
        host_path._enqueue = spy
        try:
            got = host_path.remap_host_array(
                plan, m.dst_dims, series, [1], mode='auto',
                threshold=0.2).result()
        finally:
            host_path._enqueue = real_enqueue
        assert seen == ['masked']
        assert_bitwise(got, np.ma.filled(ref, np.nan), 'sampled NaN')
        # a series "too large" to be resident whole: the NaN decision is
        # taken on the host, the data streams
        old_fraction = host_path.DEVICE_FRACTION
        host_path.DEVICE_FRACTION = 0.0
        try:
            for field, masked in ((series, True), (series[:3], False)):
                got = host_path.remap_host_array(
                    plan, m.dst_dims, field, [1], mode='auto',
                    threshold=0.2).result()
                want = oracle.remap_numpy_array(
                    csr, mm['frac_b'], m.dst_dims,
                    np.ma.masked_array(field, np.isnan(field)) if masked
                    else field, [1], 0.2 if masked else None)
                assert_bitwise(got, np.ma.filled(want, np.nan),
                               f'streamed auto, masked={masked}')
        finally:
            host_path.DEVICE_FRACTION = old_fraction
    finally:
        host_path.CHUNK_BYTES = old
        host_path.RING_SLOTS = old_ring
    # (n_a, K) fields: destination row blocks launched as their band of
    # source rows arrives, downloads under the remaining uploads
    flat = rng.standard_normal((m.n_a, 70))
    flat_holed = flat.copy()
    flat_holed[rng.random(m.n_a) < 0.2, 5:] = np.nan
    host_path.CHUNK_BYTES = 96 * 1024
    calls = []
    real_banded = host_path._banded_pipeline

    def spy(*a, **k):
        out = real_banded(*a, **k)
        calls.append(out is not None)
        return out
    host_path._banded_pipeline = spy
    try:
        for field, mode, thr in ((flat, 'fracb', None),
                                 (flat_holed, 'masked', 0.2),
                                 (flat.astype(np.float32), 'fracb', None)):
            arg = np.ma.masked_array(field, np.isnan(field)) \
                if mode == 'masked' else field
            ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims,
                                           arg, [0], thr)
            data, got_mask = host_path.remap_host_array(
                plan, m.dst_dims, field, [0], mode=mode, threshold=thr,
                want_mask=True).result()
            assert_bitwise(data, np.ma.filled(ref, np.nan),
                           f'banded {mode} {field.dtype}')
            assert np.array_equal(got_mask, np.ma.getmaskarray(ref))
        assert calls == [True, True, True]
    finally:
        host_path.CHUNK_BYTES = old
        host_path._banded_pipeline = real_banded
    # pinned results are budgeted: what is alive is accounted for, given back
    # when the arrays die, and beyond the budget results are pageable
    import gc
    del got, data
    gc.collect()
    base = host_path.pinned_bytes_alive()
    got = host_path.remap_host_array(plan, m.dst_dims, x, [1],
                                     mode='fracb').result()
    assert host_path.pinned_bytes_alive() == base + got.nbytes
    view = got[1:3]
    del got
    gc.collect()
    assert host_path.pinned_bytes_alive() == base + view.base.nbytes
    del view
    gc.collect()
    assert host_path.pinned_bytes_alive() == base
    dropped = host_path.remap_host_array(plan, m.dst_dims, x, [1],
                                         mode='fracb')
    del dropped                      # never awaited: budget returned too
    gc.collect()
    torch.cuda.synchronize()
    assert host_path.pinned_bytes_alive() == base
    old_limit = host_path.PINNED_LIMIT
    host_path.PINNED_LIMIT = 0
    try:
        ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, x, [1],
                                       None)
        got = host_path.remap_host_array(plan, m.dst_dims, x, [1],
                                         mode='fracb').result()
        assert host_path.pinned_bytes_alive() == base
        assert_bitwise(got, np.ma.filled(ref, np.nan), 'pageable result')
    finally:
        host_path.PINNED_LIMIT = old_limit
    # MaskedArray with an unmasked NaN in its data
    field = rng.standard_normal((m.n_a, 12))
    mask = rng.random((m.n_a, 12)) < 0.2
    field[5, 3] = np.nan
    mask[5, 3] = False                      # poisoned: NaN but NOT masked
    field[9, 4] = np.nan
    mask[9, 4] = True                       # ordinary: NaN under the mask
    arg = np.ma.masked_array(field, mask)
    ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, arg, [0],
                                   0.05)
    data, got_mask = host_path.remap_host_array(
        plan, m.dst_dims, field, [0], mode='masked', threshold=0.05,
        want_mask=True, host_mask=mask).result()
    ref_data = np.ma.getdata(ref)
    ref_mask = np.ma.getmaskarray(ref)
    assert np.isnan(ref_data[~ref_mask]).any()     # the NaN got through
    assert np.array_equal(got_mask, ref_mask)
    assert_bitwise(np.where(ref_mask, 0.0, data),
                   np.where(ref_mask, 0.0, ref_data), 'poisoned entry')
    # a MaskedArray that masks NOTHING: still the masked branch (the
    # normaliser is A . 1, not frac_b), and a NaN it holds goes through
    from types import SimpleNamespace
    from pyremap_amd.remapper import remap_numpy as rn
    fake = SimpleNamespace(
        _matrix=plan, engine_flags=0,
        _ds_map=SimpleNamespace(dst_grid_dims=m.dst_dims))
    clean = rng.standard_normal((m.n_a, 6))
    dirty = clean.copy()
    dirty[7, 2] = np.nan
    for values in (clean, dirty):
        arr = np.ma.masked_array(values, mask=False)
        want = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, arr,
                                        [0], 0.05)
        got = rn._remap_numpy_array(fake, arr, [0], 0.05)
        wm = np.ma.getmaskarray(want)
        assert np.array_equal(np.ma.getmaskarray(got), wm)
        assert_bitwise(np.where(wm, 0.0, np.ma.getdata(got)),
                       np.where(wm, 0.0, np.ma.getdata(want)),
                       'empty mask')
    assert np.isnan(np.ma.getdata(want)[~wm]).any()


@pytest.mark.parametrize('rows', [8, 4])
@pytest.mark.parametrize('case', ['2d', '2d_supertile', '1d', 'shard'])
def test_group_schedule_builder_matches_restatement(dev, case, rows):
    """
    remap_groups_build (the C ABI's device builder of the row-group schedule)
    against a plain-torch restatement: every array identical -- group bounds,
    union columns, member masks, the compact weights, work-slot row ids,
    frac_b, processing order.
    """
    from helpers import reference_group_schedule
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(2500, (37, 50), 2, 12, seed=31,
                                   signed=True)
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(mm['row'], mm['col'], mm['S'],
                                          mm['frac_b'], m.n_a, m.n_b,
                                          device=dev)
    dims, st = m.dst_dims, 1 << 30
    if case == '2d_supertile':
        st = 8
    elif case == '1d':
        dims = None
    elif case == 'shard':
        plan = plan.row_slice(333, 1501)       # rows 333..1500 of the grid
    ratio = plan.build_groups(dims, super_tile=st, rows=rows)
    g = plan.groups
    meta, col, mask, w, rid, frac, order, nu = reference_group_schedule(
        plan, dims, super_tile=st, rows=rows)
    assert g['union'] == nu and abs(ratio - nu / plan.nnz) < 1e-15
    assert torch.equal(g['meta'], meta)
    assert torch.equal(g['col'][:nu], col)
    assert torch.equal(g['mask'][:nu], mask)
    assert torch.equal(g['w'][:plan.nnz], w)          # bit for bit
    assert bool((g['w'][plan.nnz:] == 0).all())
    assert torch.equal(g['rid'], rid)
    assert torch.equal(g['frac'], frac)
    if order is None:
        assert plan.row_order is None
    else:
        assert torch.equal(plan.row_order, order)


def test_integration_stub_runs_on_the_c_abi_alone(dev):
    """
    The ctypes stub printed in INTEGRATION.md -- what a pyremap maintainer
    would paste -- is executed as it stands (only the library path is filled
    in): CSR from triplets, the row-group schedule from remap_groups_build,
    the apply; none of pyremap_amd's host code is involved.  Bit for bit the
    oracle, frac_b and masked branch, and the schedule's kernel really ran.
    """
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    text = open(os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'INTEGRATION.md')).read()
    # (the lower-level stub: the block that declares struct remap_apply_args)
    code = text[:text.index('class _Args(ctypes.Structure)')]
    code = text[code.rindex('```python') + len('```python'):]
    code = code[:code.index('```')]
    code = code.replace("'libremap_hip.so'", repr(engine.library_path()))
    ns = {}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    m = synthetic.conservative_map(3000, (30, 44), 1, 7, seed=12)
    mm = m.numpy()
    csr = ns['build_csr'](mm['row'], mm['col'], mm['S'], m.n_b, m.n_a)
    frac_b = torch.as_tensor(mm['frac_b'], device=dev)
    sched = ns['build_schedule'](csr, frac_b, list(m.dst_dims))
    ref_csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'],
                                m.n_b, m.n_a)
    rng = np.random.default_rng(6)
    K = 192
    x = rng.standard_normal((m.n_a, K))
    xm = x.copy()
    xm[rng.random(m.n_a) < 0.25, :] = np.nan
    for field, masked in ((x, False), (xm, True)):
        X = torch.from_numpy(field).to(dev)
        Y = torch.full((m.n_b, K), 9.0, dtype=torch.float64, device=dev)
        M = torch.zeros((m.n_b, K), dtype=torch.uint8, device=dev)
        ns['apply'](csr, sched, frac_b, X, Y, K, masked, 0.2, mask_out=M)
        ref, ref_mask = oracle.remap_flat(ref_csr, mm['frac_b'], field,
                                          masked, 0.2)
        ref[ref_mask] = np.nan
        assert_bitwise(Y.cpu().numpy(), ref, f'stub masked={masked}')
        assert np.array_equal(M.cpu().numpy().astype(bool), ref_mask)
    # a schedule that is not what it claims to be would be caught here: the
    # group kernel without the schedule's processing order gives other rows
    plain = torch.empty((m.n_b, K), dtype=torch.float64, device=dev)
    ns['apply'](csr, None, frac_b, torch.from_numpy(x).to(dev), plain, K,
                False, 0.0)
    Y2 = torch.empty((m.n_b, K), dtype=torch.float64, device=dev)
    ns['apply'](csr, sched, frac_b, torch.from_numpy(x).to(dev), Y2, K,
                False, 0.0)
    assert torch.equal(torch.nan_to_num(plain, nan=1e300),
                       torch.nan_to_num(Y2, nan=1e300))


def test_integration_plan_stub_runs_on_the_c_abi_alone(dev):
    """
    The FIRST stub of INTEGRATION.md -- the opaque plan handle, device memory
    owned by the library -- executed as printed: host triplets in, bit for
    bit the oracle out (frac_b and masked branch, conservative and bilinear
    mappings, i.e. the row-group and the LDS-patch schedule), out-of-range
    triplets refused, the handle freed.
    """
    import ctypes
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    text = open(os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'INTEGRATION.md')).read()
    code = text[text.index('```python') + len('```python'):]
    code = code[:code.index('```')]
    assert 'remap_plan_create' in code and '_Args' not in code
    code = code.replace("'libremap_hip.so'", repr(engine.library_path()))
    ns = {}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    lib = engine.load_library()
    rng = np.random.default_rng(16)
    for m, family in ((synthetic.conservative_map(3000, (30, 44), 1, 7,
                                                  seed=12), 10),
                      (synthetic.bilinear_map((13, 17), (60, 75), seed=3),
                       5)):
        mm = m.numpy()
        plan = ns['Plan'](mm['row'], mm['col'], mm['S'], mm['frac_b'],
                          m.n_b, m.n_a, list(m.dst_dims))
        info = engine._PlanInfo()
        assert lib.remap_plan_query(plan._handle, ctypes.byref(info)) == 0
        ref_csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'],
                                    m.n_b, m.n_a)
        assert (info.n_a, info.n_b, info.nnz) == (m.n_a, m.n_b,
                                                  len(ref_csr.data))
        assert info.family == family and info.device_bytes > 0
        for K in (192, 7):
            x = rng.standard_normal((m.n_a, K))
            xm = x.copy()
            xm[rng.random(m.n_a) < 0.25, :] = np.nan
            for field, masked in ((x, False), (xm, True)):
                X = torch.from_numpy(field).to(dev)
                Y = torch.full((m.n_b, K), 9.0, dtype=torch.float64,
                               device=dev)
                M = torch.zeros((m.n_b, K), dtype=torch.uint8, device=dev)
                plan.apply(X, Y, K, masked, 0.2, mask_out=M)
                ref, ref_mask = oracle.remap_flat(ref_csr, mm['frac_b'],
                                                  field, masked, 0.2)
                ref[ref_mask] = np.nan
                assert_bitwise(Y.cpu().numpy(), ref,
                               f'plan stub family {family} K={K} '
                               f'masked={masked}')
                assert np.array_equal(M.cpu().numpy().astype(bool),
                                      ref_mask)
        torch.cuda.synchronize()
        del plan                       # remap_plan_destroy
    bad_row = mm['row'].copy()
    bad_row[5] = m.n_b + 3
    with pytest.raises(RuntimeError, match='outside'):
        ns['Plan'](bad_row, mm['col'], mm['S'], mm['frac_b'], m.n_b, m.n_a,
                   list(m.dst_dims))
    with pytest.raises(RuntimeError, match='destination grid'):
        ns['Plan'](mm['row'], mm['col'], mm['S'], mm['frac_b'], m.n_b,
                   m.n_a, [3, 5])


@pytest.mark.parametrize('case', ['2d', '1d', 'shard', 'bilinear'])
def test_patch_plan_builder_matches_restatement(dev, case):
    """remap_patches_build (device builder of the LDS patch plan) against a
    plain-torch restatement: every array and statistic identical."""
    from helpers import reference_patch_plan
    from pyremap_amd import engine, synthetic
    if case == 'bilinear':
        m = synthetic.bilinear_map((17, 23), (61, 90), seed=2)
    else:
        m = synthetic.conservative_map(2500, (37, 50), 1, 7, seed=32)
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(mm['row'], mm['col'], mm['S'],
                                          mm['frac_b'], m.n_a, m.n_b,
                                          device=dev)
    dims, tile = m.dst_dims, (4, 8)
    if case == '1d':
        dims, tile = None, (1, 32)
    elif case == 'shard':
        plan = plan.row_slice(211, 1403)
    ratio = plan.build_patches(dims, tile=tile, lds_budget=150 * 1024)
    p = plan.patches
    assert p['tile'] == tile          # fits as asked: same tile both sides
    ptr, ucol, prow, lidx, val, order, distinct, umax, emax = \
        reference_patch_plan(plan, dims, tile)
    assert (p['distinct'], p['umax'], p['emax']) == (distinct, umax, emax)
    assert abs(ratio - distinct / plan.nnz) < 1e-15
    assert torch.equal(p['ptr'], ptr)
    assert torch.equal(p['ucol'], ucol)
    assert torch.equal(p['rowptr'], prow)
    assert torch.equal(p['lidx'], lidx)
    assert torch.equal(p['val'], val)
    if order is None:
        assert plan.row_order is None
    else:
        assert torch.equal(plan.row_order, order)
    # a budget that forces smaller tiles still yields a consistent plan
    plan.build_patches(dims, tile=tile, lds_budget=24 * 1024)
    q = plan.patches
    if q is not None:
        ref = reference_patch_plan(plan, dims, q['tile'])
        assert torch.equal(q['lidx'], ref[3]) and q['umax'] == ref[7]


def test_bench_multi_rank_line_and_a_hung_exchange(tmp_path):
    """
    `bench.py --gpus 2` as the driver launches it (torch.distributed.run, one
    process per rank; here gloo, both ranks on this GPU): the JSON line
    carries the sharded metric and the exchange timings.  When the optional
    exchange measurements never return (BENCH_TEST_HANG) the watchdog still
    gets the metric line out -- with "status": "exchange_hung" and a NON-ZERO
    exit code: a hung exchange must not look like a clean run.
    """
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
            '--nproc-per-node', '2', '--master-addr', '127.0.0.1']
    bench = [os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '5',
             '--warmup', '2', '--backend', 'gloo']
    # the third pass is plain `python bench.py --gpus 2 ...` with no launcher
    # around it: bench.py starts its ranks itself (as a child) and relays
    # rank 0's line -- with the extras of an N > 1 run: the sharded headline,
    # config 4 and config 5 (the workloads BASELINE.json names for 8 GPUs)
    for port, env_extra in (('29571', {}),
                            ('29572', {'BENCH_TEST_HANG': '1',
                                       'BENCH_OPTIONAL_TIMEOUT_S': '5'}),
                            ('29573', {'BENCH_TEST_HANG_BIG': '1',
                                       'BENCH_BIG_TIMEOUT_S': '5'}),
                            (None, {})):
        env = dict(os.environ, **env_extra)
        for name in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
            env.pop(name, None)
        cmd = [sys.executable] + bench if port is None else \
            base + ['--master-port', port] + bench + ['--no-extra']
        proc = subprocess.run(cmd, capture_output=True, text=True, env=env,
                              timeout=900, cwd=str(tmp_path))
        hung = 'BENCH_TEST_HANG' in env_extra
        big_hung = 'BENCH_TEST_HANG_BIG' in env_extra
        # (an extra workload that does not return costs its rows, not the
        # record: the line is printed, status says so -- and the exit code
        # is 4, not that of a clean run; 3 = the metric's own exchange hung)
        assert (proc.returncode != 0) == (hung or big_hung), \
            proc.stderr[-2000:]
        last = proc.stdout.strip().splitlines()[-1]
        assert len(last) < 4096
        line = json.loads(last)
        assert line['status'] == ('exchange_hung' if hung else
                                  'extras_timed_out' if big_hung else 'ok')
        assert line['n_gpus'] == 2 and line['scaling'] == 'strong'
        assert line['value'] > 0 and 0 < line['roofline']['frac'] < 1
        multi = line['multi_gpu']
        assert multi['broadcast_ms'] > 0 and multi['kernel_phase_ms'] > 0
        assert multi['ranks'] == 2 and multi['backend'] == 'gloo'
        # the source mesh is numbered as MPAS numbers its cells, and each
        # rank still needs only about half of the source rows
        assert line['config']['locality'] == 'mesh'
        assert multi['packed_fraction_of_broadcast'] < 0.8
        assert ('optional_measurements' in multi) == (hung or big_hung)
        if port is None:
            # [kernel-phase ms of the slowest rank, fraction of 2 x 8 TB/s,
            # packed fraction of the source rows a rank holds]
            rows = multi['workloads']
            assert set(rows) == {'headline', 'config4', 'config5'}, rows
            for tag, row in rows.items():
                assert len(row) == 3 and row[0] > 0, (tag, row)
                assert 0 < row[1] < 1 and 0 < row[2] <= 1, (tag, row)
        # everything else is in the side file the line names
        details = json.load(open(os.path.join(repo, line['details'])))
        assert details['line']['value'] == line['value']
        assert details['schedule']['family']


def test_bench_single_gpu_line_parses_and_is_small(tmp_path):
    """
    The driver's command (`python bench.py --gpus 1 --steps K --warmup W`):
    the last stdout line is JSON under 4 KB with `roofline` and
    `cpu_baseline` in it (round 3's had outgrown the driver's 8 KB tail).
    """
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    side = str(tmp_path / 'extra.json')
    proc = subprocess.run(
        [sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '1',
         '--steps', '5', '--warmup', '2', '--no-extra', '--cpu-seconds', '3',
         '--details', side],
        capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert proc.returncode == 0, proc.stderr[-2000:]
    last = proc.stdout.strip().splitlines()[-1]
    assert len(last) < 4096
    line = json.loads(last)
    assert line['n_gpus'] == 1 and line['steps'] == 5 and line['warmup'] == 2
    assert line['status'] == 'ok' and line['multi_gpu'] is None
    roof, cpu = line['roofline'], line['cpu_baseline']
    assert roof['bound'] == 'hbm' and 0.2 < roof['frac'] < 1
    assert abs(roof['achieved'] - roof['bytes_alg_per_launch'] /
               (roof['kernel_ms_mean'] * 1e-3) / 1e9) < 1e-3 * roof['achieved']
    assert cpu['cores'] == 1 and cpu['value'] > 0 and cpu['port_value'] > 0
    assert cpu['kind'] in ('reference', 'port')
    # value = the units of the timed steps over their wall time
    units = line['config']['n_b'] * line['config']['fields_K']
    assert abs(line['value'] - units / (line['ms_per_step'] * 1e-3)) < \
        1e-6 * line['value']
    details = json.load(open(side))
    assert details['line'] == line and details['phases_s']['total_s'] > 0


def test_plan_handle_device_inputs_and_strided_fields(dev):
    """
    The opaque plan handle beyond what INTEGRATION.md's stub uses: triplets
    given as DEVICE arrays (host_input = 0), a `(Time, nCells, levels)` field
    addressed in place through remap_field's strides, float32 input, the
    masked branch with a byte mask -- against the oracle, bit for bit.
    """
    import ctypes
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    lib = engine.load_library()
    m = synthetic.conservative_map(4000, (36, 50), 1, 7, seed=44)
    mm = m.numpy()
    dims = (ctypes.c_int64 * 2)(*m.dst_dims)
    row = torch.as_tensor(mm['row'], dtype=torch.int32, device=dev)
    col = torch.as_tensor(mm['col'], dtype=torch.int32, device=dev)
    S = torch.as_tensor(mm['S'], dtype=torch.float64, device=dev)
    fb = torch.as_tensor(mm['frac_b'], dtype=torch.float64, device=dev)
    handle = ctypes.c_void_p()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.remap_plan_create(
        m.n_b, m.n_a, S.numel(), row.data_ptr(), col.data_ptr(),
        S.data_ptr(), 1, fb.data_ptr(), 0, dims, 2, stream,
        ctypes.byref(handle))
    assert rc == 0, lib.remap_last_error()
    del row, col, S, fb                       # not retained by the plan
    try:
        csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                                m.n_a)
        rng = np.random.default_rng(3)
        T, L = 3, 60
        for dtype in (np.float64, np.float32):
            field = rng.standard_normal((T, m.n_a, L)).astype(dtype)
            field[:, rng.random(m.n_a) < 0.2, 30:] = np.nan
            for masked in (False, True):
                arg = np.ma.masked_array(field, np.isnan(field)) if masked \
                    else field
                ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims,
                                               arg, [1],
                                               0.1 if masked else None)
                X = torch.from_numpy(field).to(dev)
                Y = torch.full((T, m.n_b, L), 5.0, dtype=torch.float64,
                               device=dev)
                M = torch.zeros((T, m.n_b, L), dtype=torch.uint8, device=dev)
                f = engine._Field()
                f.X, f.Y = X.data_ptr(), Y.data_ptr()
                f.x_dtype = engine.DTYPE_F64 if dtype == np.float64 \
                    else engine.DTYPE_F32
                f.mode = engine.MODE_MASKED if masked else engine.MODE_FRACB
                f.n_batch, f.k_inner = T, L
                f.x_row_stride, f.x_batch_stride = L, m.n_a * L
                f.y_row_stride, f.y_batch_stride = L, m.n_b * L
                f.threshold = 0.1
                f.mask_out = M.data_ptr()
                rc = lib.remap_plan_apply(handle, ctypes.byref(f), stream)
                assert rc == 0, lib.remap_last_error()
                got = Y.cpu().numpy().reshape((T,) + tuple(m.dst_dims) + (L,))
                want = np.ma.filled(np.ma.masked_array(ref).astype(
                    np.float64), np.nan)
                what = f'plan handle {np.dtype(dtype).name} masked={masked}'
                assert_bitwise(got, want, what)
                assert np.array_equal(
                    M.cpu().numpy().astype(bool).reshape(got.shape),
                    np.ma.getmaskarray(np.ma.masked_array(ref))), what
        torch.cuda.synchronize()
    finally:
        lib.remap_plan_destroy(handle)


def test_plan_handle_time_ncells(dev):
    """
    The opaque handle on the reference's most common layout, (Time, nCells):
    in place (n_batch = Time, k_inner = 1) through `spmm_rowcell` as created,
    through the LDS-staged `spmm_patchcell` once
    remap_plan_prepare_short_runs has built its patch plan (a second call is
    a no-op; the plan's device_bytes grows once) -- same bits either way.
    """
    import ctypes
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    lib = engine.load_library()
    m = synthetic.conservative_map(5000, (40, 60), 1, 7, seed=45,
                                   locality='mesh')
    mm = m.numpy()
    dims = (ctypes.c_int64 * 2)(*m.dst_dims)
    handle = ctypes.c_void_p()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def ptr(a):
        return a.ctypes.data_as(ctypes.c_void_p)
    row = np.ascontiguousarray(mm['row'], dtype=np.int32)
    col = np.ascontiguousarray(mm['col'], dtype=np.int32)
    S = np.ascontiguousarray(mm['S'])
    fb = np.ascontiguousarray(mm['frac_b'])
    rc = lib.remap_plan_create(m.n_b, m.n_a, S.size, ptr(row), ptr(col),
                               ptr(S), 1, ptr(fb), 1, dims, 2, stream,
                               ctypes.byref(handle))
    assert rc == 0, lib.remap_last_error()
    try:
        csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                                m.n_a)
        rng = np.random.default_rng(8)
        T = 90
        field = rng.standard_normal((T, m.n_a))
        field[:, rng.random(m.n_a) < 0.2] = np.nan
        ref = np.ma.filled(oracle.remap_numpy_array(
            csr, mm['frac_b'], m.dst_dims,
            np.ma.masked_array(field, np.isnan(field)), [1], 0.1), np.nan)
        X = torch.from_numpy(field).to(dev)
        info = engine._PlanInfo()
        sizes = []
        for prepared in (False, True, True):
            if prepared:
                rc = lib.remap_plan_prepare_short_runs(handle, stream)
                assert rc == 0, lib.remap_last_error()
            assert lib.remap_plan_query(handle, ctypes.byref(info)) == 0
            sizes.append(int(info.device_bytes))
            Y = torch.full((T, m.n_b), 5.0, dtype=torch.float64, device=dev)
            f = engine._Field()
            f.X, f.Y = X.data_ptr(), Y.data_ptr()
            f.x_dtype, f.mode = engine.DTYPE_F64, engine.MODE_MASKED
            f.n_batch, f.k_inner = T, 1
            f.x_row_stride, f.x_batch_stride = 1, m.n_a
            f.y_row_stride, f.y_batch_stride = 1, m.n_b
            f.threshold = 0.1
            rc = lib.remap_plan_apply(handle, ctypes.byref(f), stream)
            assert rc == 0, lib.remap_last_error()
            assert_bitwise(Y.cpu().numpy().reshape((T,) + m.dst_dims), ref,
                           f'handle (Time, nCells) prepared={prepared}')
        assert sizes[1] > sizes[0] and sizes[2] == sizes[1]
    finally:
        lib.remap_plan_destroy(handle)


def test_c_abi_from_plain_c(tmp_path):
    """examples/c_abi_plan.c -- gcc, no Python, no C++: the plan handle from
    C, every value compared in C with a sequential multiply-then-add."""
    import shutil
    import subprocess
    from pyremap_amd import engine
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which('gcc')
    if gcc is None or not os.path.exists('/opt/rocm/lib/libamdhip64.so'):
        pytest.skip('needs gcc and the HIP runtime library')
    lib_dir = os.path.dirname(engine.library_path())
    exe = str(tmp_path / 'c_abi_plan')
    subprocess.run(
        [gcc, '-O2', '-ffp-contract=off', '-D__HIP_PLATFORM_AMD__',
         '-I/opt/rocm/include', '-I' + os.path.join(repo, 'include'),
         os.path.join(repo, 'examples', 'c_abi_plan.c'), '-o', exe,
         '-L' + lib_dir, '-lremap_hip', '-L/opt/rocm/lib', '-lamdhip64',
         '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,/opt/rocm/lib', '-lm'],
        check=True, capture_output=True, text=True)
    proc = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert ' 0 differ' in proc.stdout and 'gfx950' in proc.stdout
    assert 'nCells) in place: 0 differ' in proc.stdout
