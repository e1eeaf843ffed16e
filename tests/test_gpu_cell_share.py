"""
The shared form of kernel family 10 in the masked mode with whole cells
missing (round 6, csrc/spmm_cellshare.h): REMAP_FLAG_CELL_MASKS on a plan
with the shared lists (`remap_schedule_auto` builds them on entry-rich
mappings; `tune[5]` 0 or 32) -- one normaliser per ROW, the union of a 4 x 8
tile through the LDS ring.  Every value against the oracle, bit for bit, through
the C ABI, whatever is missing: a wave that meets a cell missing in some
columns only, or a NaN / Inf weight on a missing cell, redoes its group with
per-element normalisers inside the launch.  Reference arithmetic:
remap_numpy.py:262-266, 277-278.
"""
import numpy as np
import pytest

from helpers import assert_bitwise
from test_gpu_group_forms import _check, _fields

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

SHARE = [10, 0, 2, 0, 3, 32]


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from pyremap_amd import engine
    engine.load_library()
    return torch.device('cuda', 0)


def _problem(dev, n_a=1500, dims=(38, 60), k=(6, 22), seed=5, S_edit=None):
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(n_a, dims, k[0], k[1], seed=seed,
                                   signed=True, locality='mesh')
    mm = m.numpy()
    S = mm['S'].copy()
    if S_edit:
        S_edit(S)
    plan = engine.RemapPlan.from_triplets(
        mm['row'], mm['col'], S, mm['frac_b'], m.n_a, m.n_b, index_base=1,
        device=dev)
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, S, m.n_b, m.n_a)
    plan.build_groups(m.dst_dims, rows=8, share=4)
    return m, mm, plan, csr


@pytest.mark.parametrize('K', [130, 192, 256, 300, 1024])
def test_cell_share_bitwise(dev, K):
    """Nothing missing / whole cells / single values / cells and levels; K
    tails; the three work-list orders; with the flag (the shared form), and
    the same bits without it and on the 8-row groups (tune[5] = 0)."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev)
    for tag, x in _fields(m.n_a, K, K + 9):
        for order in (3, 2, 1):
            tune = [10, 0, 2, 0, order, 32]
            _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_MASKED, 0.3,
                   tune, f'cell share {tag} K={K} tune={tune}',
                   flags=engine.FLAG_CELL_MASKS | engine.FLAG_TUNE_HINT)
        # tune[5] = 0: the plan's own choice (the shared form again);
        # 8: the 8-row groups (spmm_groupmask.h); 9: per-lane normalisers
        for t5 in (0, 8, 9):
            _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_MASKED, 0.3,
                   [10, 1, 2, 1, 3, t5], f'{tag} K={K} tune[5]={t5}',
                   flags=engine.FLAG_CELL_MASKS)


def test_cell_share_runs_the_shared_kernel(dev):
    """The dispatch: with the flag and tune[5] = 32 the call is served
    without REMAP_FLAG_TUNE_HINT (the shared form of the frac_b mode declines
    the masked mode: an error without the hint); threshold 0 and a threshold
    above every row's den."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev)
    tag, x = _fields(m.n_a, 512, 3)[1]
    for thr in (0.0, 0.3, 50.0):
        _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_MASKED, thr,
               SHARE, f'{tag} thr={thr}', flags=engine.FLAG_CELL_MASKS)
    xd = torch.from_numpy(x).to(dev)
    with pytest.raises(engine.EngineError, match='shared form'):
        engine.remap_tensor(plan, None, xd, [0], engine.MODE_MASKED,
                            threshold=0.3, tune=SHARE)


def test_cell_share_long_lists_and_one_dimensional(dev):
    """Lists of more than 128 union entries (a second segment), a 1-D
    destination whose last supergroup is partial."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, n_a=700, dims=(30, 44), k=(24, 60),
                                seed=11)
    longest = int(np.diff(
        plan.groups['share']['meta'][:, 0].cpu().numpy()).max())
    assert longest > 128, longest
    for tag, x in _fields(m.n_a, 384, 5):
        _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_MASKED, 0.2,
               SHARE, f'long lists {tag}', flags=engine.FLAG_CELL_MASKS)
    # the rows taken as a 1-D destination: supergroups of 32 consecutive
    # rows, the last one partial (30 * 44 = 1 320 = 41 * 32 + 8)
    plan.build_groups(None, rows=8, share=4)
    assert m.n_b % 32 != 0
    for tag, x in _fields(m.n_a, 256, 6):
        _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_MASKED, 0.2,
               SHARE, f'1-D {tag}', flags=engine.FLAG_CELL_MASKS)


def test_cell_share_odd_weights_take_the_general_form(dev):
    """A NaN or an Inf weight on a cell missing in every column: `a * 0.0` is
    NaN there, the skip of the fast form does not apply."""
    from pyremap_amd import engine

    def edit(S):
        S[7] = np.inf
        S[1000] = np.nan
        S[2000] = -np.inf
    m, mm, plan, csr = _problem(dev, n_a=900, dims=(24, 40), k=(6, 18),
                                seed=9, S_edit=edit)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((m.n_a, 256))
    x[rng.random(m.n_a) < 0.3] = np.nan
    for c in (mm['col'][7], mm['col'][1000], mm['col'][2000]):
        x[c - 1] = np.nan
    _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_MASKED, 0.1, SHARE,
           'odd weights', flags=engine.FLAG_CELL_MASKS)


def test_cell_share_fma_is_close(dev):
    """REMAP_FLAG_FMA: rtol 1e-12, same mask."""
    from oracle import oracle
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev)
    for tag, x in _fields(m.n_a, 512, 77):
        xd = torch.from_numpy(x).to(dev)
        y, mask = engine.remap_tensor(
            plan, None, xd, [0], engine.MODE_MASKED, threshold=0.3,
            want_mask=True, tune=SHARE,
            flags=engine.FLAG_CELL_MASKS | engine.FLAG_FMA)
        ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, True, 0.3)
        assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask), tag
        ok = ~ref_mask
        np.testing.assert_allclose(y.cpu().numpy()[ok], ref[ok], rtol=1e-12,
                                   atol=1e-13)


def _reference_tnl(csr, frac_b, f, thr):
    from oracle import oracle
    T, n_a, L = f.shape
    flat = np.ascontiguousarray(f.transpose(1, 0, 2)).reshape(n_a, T * L)
    ref, ref_mask = oracle.remap_flat(csr, frac_b, flat, True, thr)
    ref = ref.copy()
    ref[ref_mask] = np.nan
    return ref.reshape(ref.shape[0], T, L).transpose(1, 0, 2)


@pytest.mark.parametrize('shape', [(8, 64), (5, 60), (3, 100), (2, 130),
                                   (40, 7), (1, 192)])
def test_cell_share_time_cells_levels_in_place(dev, shape):
    """(Time, nCells, nVertLevels) read in place -- level runs shorter and
    longer than a K tile, batch-aligned tiles, one and two batches -- with
    land cells (the fast form), with bathymetry (every group redone), with a
    mask that changes in time."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev)
    T, L = shape
    rng = np.random.default_rng(T * 1000 + L)
    base = rng.standard_normal((T, m.n_a, L))
    land = base.copy()
    land[:, rng.random(m.n_a) < 0.25, :] = np.nan
    land[:, 0, :] = np.nan
    depth = rng.integers(1, L + 1, m.n_a)
    bath = land.copy()
    bath[:, np.arange(L)[None, :] >= depth[:, None]] = np.nan
    varying = land.copy()
    varying[T // 2, 11, L // 2] = np.nan
    for tag, f in (('no NaN', base), ('land', land), ('bathymetry', bath),
                   ('one value', varying)):
        ref = _reference_tnl(csr, mm['frac_b'], f, 0.3)
        fd = torch.from_numpy(f).to(dev)
        y, mask = engine.remap_tensor(
            plan, None, fd, [1], engine.MODE_MASKED, threshold=0.3,
            flags=engine.FLAG_CELL_MASKS | engine.FLAG_TUNE_HINT, tune=SHARE,
            want_mask=True)     # (7 levels: no 16-byte pieces, handed on)
        what = f'{tag} (T={T}, L={L})'
        assert_bitwise(y.cpu().numpy(), ref, what)
        assert np.array_equal(mask.cpu().numpy().astype(bool),
                              np.isnan(ref)), what


def test_cell_share_batches_further_apart_than_32_bit_offsets(dev):
    """Time slices 2 GiB apart (a 3.7 M-cell mesh): the shared form addresses
    X with flat 64-bit addresses, in the ring and in the redone groups."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev)
    T, L = 4, 64
    stride = (1 << 28) + 64           # elements: 2 GiB and a bit per slice
    rng = np.random.default_rng(33)
    f = rng.standard_normal((T, m.n_a, L))
    f[:, rng.random(m.n_a) < 0.25, :] = np.nan
    fb = f.copy()
    depth = rng.integers(1, L + 1, m.n_a)
    fb[:, np.arange(L)[None, :] >= depth[:, None]] = np.nan
    big = torch.zeros(T * stride, dtype=torch.float64, device=dev)
    y = torch.empty((T, m.n_b, L), dtype=torch.float64, device=dev)
    for tag, field in (('land', f), ('bathymetry', fb)):
        view = torch.as_strided(big, (T, m.n_a, L), (stride, L, 1))
        view.copy_(torch.from_numpy(field).to(dev))
        y.fill_(7.0)
        engine.apply_strided(
            plan, big, y, n_batch=T, k_inner=L, x_row_stride=L,
            x_batch_stride=stride, y_row_stride=L, y_batch_stride=m.n_b * L,
            mode=engine.MODE_MASKED, threshold=0.3,
            flags=engine.FLAG_CELL_MASKS, tune=SHARE)
        assert_bitwise(y.cpu().numpy(),
                       _reference_tnl(csr, mm['frac_b'], field, 0.3),
                       f'{tag}, wide batches')
