"""
The shared form of kernel family 10 (round 6, csrc/spmm_groupshare.h): 2 or 4
waves, one union of source rows per 16 / 32 destination rows, sent once into
an LDS ring by LDS-DMA -- every value against the oracle, bit for bit,
through the C ABI; the shared lists (`remap_share_build`) against a numpy
restatement.  Reference arithmetic: remap_numpy.py:258-278 (each row adds its
own entries in ascending column order, scipy's csr_matvecs).
"""
import numpy as np
import pytest

from helpers import assert_bitwise
from test_gpu_group_forms import _check, _fields

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from pyremap_amd import engine
    engine.load_library()
    return torch.device('cuda', 0)


def _problem(dev, share, n_a=1500, dims=(38, 60), k=(6, 22), seed=5,
             two_d=True):
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
    plan.build_groups(m.dst_dims if two_d else None, rows=8, share=share)
    return m, mm, plan, csr


@pytest.mark.parametrize('two_d', [True, False])
@pytest.mark.parametrize('share', [2, 4])
def test_shared_lists_match_a_numpy_restatement(dev, share, two_d):
    """share_col / share_mask / share_meta: per supergroup of 8 * share work
    slots the sorted distinct columns of its rows and who owns them; with a
    2-D grid a supergroup is a 4 x 4 / 4 x 8 tile (whole tiles inside)."""
    m, mm, plan, csr = _problem(dev, share, two_d=two_d)
    indptr, indices = csr.indptr, csr.indices
    g = plan.groups
    sh = g['share']
    assert sh['waves'] == share
    rid = g['rid'].cpu().numpy()
    meta = sh['meta'].cpu().numpy()
    col = sh['col'].cpu().numpy()
    mask = sh['mask'].cpu().numpy().view(np.uint32)
    SR = 8 * share
    n_super = (m.n_b + SR - 1) // SR
    assert meta.shape == (n_super + 1, 2)
    total = 0
    for s in range(n_super):
        slots = range(s * SR, min((s + 1) * SR, m.n_b))
        want = {}
        for member, slot in enumerate(slots):
            r = rid[slot]
            for c in indices[indptr[r]:indptr[r + 1]]:
                want[int(c)] = want.get(int(c), 0) | (1 << member)
        lo, hi = meta[s, 0], meta[s + 1, 0]
        assert lo == total
        cols = sorted(want)
        assert hi - lo == len(cols), s
        assert col[lo:hi].tolist() == cols, s
        assert mask[lo:hi].tolist() == [want[c] for c in cols], s
        total += len(cols)
    assert total == sh['union']
    # readable zeros behind the lists
    assert not col[total:total + 256].any()
    assert not mask[total:total + 256].any()
    assert abs(sh['ratio'] - total / plan.nnz) < 1e-12
    if two_d:
        # a supergroup whose tile lies inside the grid is that tile
        my, mx = m.dst_dims
        ty, tx = 4, 2 * share
        r0 = rid[:SR]
        assert sorted(r0.tolist()) == sorted(
            y * mx + x for y in range(ty) for x in range(tx))


@pytest.mark.parametrize('K', [130, 192, 256, 300, 1024])
def test_shared_form_bitwise(dev, K):
    """The frac_b and raw modes, one or two K tiles per wave, the work-list
    orders; NaNs in the field propagate as the reference lets them
    (remap_numpy.py:268: raw data through `matrix.dot`)."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, 4)
    fields = _fields(m.n_a, K, K + 4)
    for tiles in (1, 2):
        for order in (3, 2, 1):
            tune = [10, 0, tiles, 0, order, 32]
            for mode in (engine.MODE_FRACB, engine.MODE_RAW):
                for tag, x in fields:
                    if tag not in ('no NaN', 'single values'):
                        continue
                    _check(plan, csr, mm['frac_b'], x, dev, mode, 0.0, tune,
                           f'share K={K} {tag} tune={tune} mode={mode}')


@pytest.mark.parametrize('K', [104, 112, 128])
def test_shared_form_one_field_of_many_levels(dev, K):
    """At most 128 columns -- ONE 3-D field of 104 ... 128 levels: the shared
    form with one K tile per wave, whatever tune[2] says; below 104 columns
    the call is declined (the 8-row groups are faster there) or, under
    REMAP_FLAG_TUNE_HINT, handed to them."""
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, 4)
    fields = _fields(m.n_a, K, K + 1)
    for tune in ([10, 0, 2, 0, 3, 32], [10, 0, 1, 0, 2, 32]):
        for mode in (engine.MODE_FRACB, engine.MODE_RAW):
            for tag, x in fields[:1] + fields[2:3]:
                _check(plan, csr, mm['frac_b'], x, dev, mode, 0.0, tune,
                       f'one tile K={K} {tag} tune={tune} mode={mode}')
    x = fields[0][1][:, :100].copy()
    xd = torch.from_numpy(x).to(dev)
    with pytest.raises(engine.EngineError, match='shared form'):
        engine.remap_tensor(plan, None, xd, [0], engine.MODE_FRACB,
                            tune=[10, 0, 2, 0, 3, 32])
    _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_FRACB, 0.0,
           [10, 0, 2, 0, 3, 32], 'K=100 under the hint',
           flags=engine.FLAG_TUNE_HINT)


@pytest.mark.parametrize('K', [34, 48, 60, 64])
def test_narrow_shared_form_bitwise(dev, K):
    """At most 64 columns -- ONE 3-D field of up to 64 levels: a lane per
    column, two union entries per LDS-DMA instruction
    (csrc/spmm_narrowshare.h); the frac_b and raw modes, the work-list
    orders, NaNs propagating as the reference lets them; as (n, K) and as
    batches of short level runs read in place."""
    from oracle import oracle
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, 4)
    fields = _fields(m.n_a, K, K + 2)
    for order in (3, 2, 1):
        tune = [10, 0, 2, 0, order, 32]
        for mode in (engine.MODE_FRACB, engine.MODE_RAW):
            for tag, x in fields[:1] + fields[2:3]:
                _check(plan, csr, mm['frac_b'], x, dev, mode, 0.0, tune,
                       f'narrow K={K} {tag} tune={tune} mode={mode}')
    # (T, n, L) in place: K = T * L columns in runs of L
    # (runs of 16 levels: flat tiles, the narrow form; of 18: batch-aligned
    # tiles, which it declines -- handed to the 8-row groups under the hint)
    for T, L in ((4, 16), (3, 16), (2, 18)):
        rng = np.random.default_rng(K + T)
        f = rng.standard_normal((T, m.n_a, L))
        flat = np.ascontiguousarray(f.transpose(1, 0, 2)).reshape(m.n_a,
                                                                  T * L)
        ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], flat, False,
                                          0.0)
        ref = ref.copy()
        ref[ref_mask] = np.nan
        y = engine.remap_tensor(plan, None, torch.from_numpy(f).to(dev), [1],
                                engine.MODE_FRACB, tune=[10, 0, 2, 0, 3, 32],
                                flags=engine.FLAG_TUNE_HINT if L == 18 else 0)
        assert_bitwise(y.cpu().numpy(),
                       ref.reshape(m.n_b, T, L).transpose(1, 0, 2),
                       f'narrow (T={T}, n, L={L})')


def test_shared_form_long_lists_one_dimensional_and_fma(dev):
    """Lists of more than 128 union entries (a second segment of lane-held
    columns and masks), a 1-D destination (supergroups of consecutive rows,
    the last one partial), REMAP_FLAG_FMA at rtol 1e-13."""
    from oracle import oracle
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, 4, n_a=700, dims=(30, 44), k=(24, 60),
                                seed=11)
    meta = plan.groups['share']['meta'][:, 0].cpu().numpy()
    longest = int(np.diff(meta).max())
    assert longest > 128, longest   # segments of 128
    rng = np.random.default_rng(3)
    x = rng.standard_normal((m.n_a, 384))
    for tune in ([10, 0, 2, 0, 3, 32], [10, 0, 1, 0, 2, 32]):
        _check(plan, csr, mm['frac_b'], x, dev, engine.MODE_FRACB, 0.0, tune,
               f'long lists, tune={tune}')
    # 1-D destination, n_b no multiple of the supergroup
    m1, mm1, plan1, csr1 = _problem(dev, 4, n_a=900, dims=(1, 1013),
                                    k=(5, 14), seed=2, two_d=False)
    x1 = rng.standard_normal((m1.n_a, 258))
    _check(plan1, csr1, mm1['frac_b'], x1, dev, engine.MODE_FRACB, 0.0,
           [10, 0, 2, 0, 3, 32], '1-D')
    # fused multiply-add: opt-in, close
    xd = torch.from_numpy(x).to(dev)
    y = engine.remap_tensor(plan, None, xd, [0], engine.MODE_FRACB,
                            tune=[10, 0, 2, 0, 3, 32],
                            flags=engine.FLAG_FMA)
    ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, False, 0.0)
    ok = ~ref_mask
    np.testing.assert_allclose(y.cpu().numpy()[ok], ref[ok], rtol=1e-12,
                               atol=1e-13)


def test_shared_form_layouts_and_what_it_declines(dev):
    """(Time, nCells, L) in place (batch strides, 256 levels and 2 x 96);
    float32 fields, few columns, odd strides, the masked mode and lists of
    two groups are declined -- an error when demanded, the 8-row groups of
    the same schedule under REMAP_FLAG_TUNE_HINT (same bits)."""
    from oracle import oracle
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, 4)
    rng = np.random.default_rng(8)
    tune = [10, 0, 2, 2, 3, 32]
    for T, L in ((2, 256), (3, 96)):
        f = rng.standard_normal((T, m.n_a, L))
        fd = torch.from_numpy(f).to(dev)
        y = engine.remap_tensor(plan, None, fd, [1], engine.MODE_FRACB,
                                tune=tune)
        flat = np.ascontiguousarray(f.transpose(1, 0, 2)).reshape(m.n_a, -1)
        ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], flat, False, 0.0)
        ref = ref.copy()
        ref[ref_mask] = np.nan
        got = y.cpu().numpy().transpose(1, 0, 2).reshape(m.n_b, -1)
        assert_bitwise(got, ref, f'(T={T}, n, L={L})')
    x = rng.standard_normal((m.n_a, 96))
    x32 = rng.standard_normal((m.n_a, 256)).astype(np.float32)
    xodd = rng.standard_normal((m.n_a, 257))
    for what, field in (('96 columns', x), ('float32', x32),
                        ('odd stride', xodd)):
        fd = torch.from_numpy(field).to(dev)
        with pytest.raises(engine.EngineError, match='shared form'):
            engine.remap_tensor(plan, None, fd, [0], engine.MODE_FRACB,
                                tune=tune)
        _check(plan, csr, mm['frac_b'], field, dev, engine.MODE_FRACB, 0.0,
               tune, what, flags=engine.FLAG_TUNE_HINT)
    # the masked mode keeps the 8-row groups (its shared form is
    # spmm_timeshare, under REMAP_FLAG_BATCH_MASKS); lists of two groups have
    # no kernel
    holed = rng.standard_normal((m.n_a, 256))
    holed[rng.random(m.n_a) < 0.2] = np.nan
    with pytest.raises(engine.EngineError, match='shared form'):
        engine.remap_tensor(plan, None, torch.from_numpy(holed).to(dev), [0],
                            engine.MODE_MASKED, threshold=0.3, tune=tune)
    _check(plan, csr, mm['frac_b'], holed, dev, engine.MODE_MASKED, 0.3,
           tune, 'masked, hint', flags=engine.FLAG_TUNE_HINT)
    m2, mm2, plan2, csr2 = _problem(dev, 2)
    with pytest.raises(engine.EngineError, match='shared form'):
        engine.remap_tensor(plan2, None, torch.from_numpy(
            rng.standard_normal((m2.n_a, 256))).to(dev), [0],
            engine.MODE_FRACB, tune=tune)
    # without the lists the switch is an error as well
    plan.build_groups(m.dst_dims, super_tile=32, rows=8)
    with pytest.raises(engine.EngineError, match='shared form'):
        engine.remap_tensor(plan, None, torch.from_numpy(
            rng.standard_normal((m.n_a, 256))).to(dev), [0],
            engine.MODE_FRACB, tune=tune)


def test_fracb_division_fast_path_and_its_guards(dev):
    """The frac_b mode divides a row by ONE wave-uniform number: for ordinary
    operands the epilogue refines 1 / frac_b once per row and finishes each
    element with a multiply and two FMAs -- the very instructions the full
    IEEE division sequence ends with when its scaling steps scale nothing
    (csrc/spmm_device.h: finish_row) -- and anything else (zeros, denormals,
    1e-250, 1e300, Inf, NaN, a frac_b outside [2^-126, 2^126]) takes the full
    sequence.  Every value equals the C division of the oracle, bit for bit
    (`num[ok] /= den[ok]`, remap_numpy.py:277), in every kernel family that
    shares the epilogue."""
    from oracle import oracle
    from pyremap_amd import engine
    rng = np.random.default_rng(12)
    n_cells, K = 32, 256
    x = rng.standard_normal((n_cells, K)) * \
        10.0 ** rng.integers(-200, 150, (n_cells, 1))
    specials = [0.0, -0.0, 1e-310, -4e-320, 1e-250, 2.0 ** -800,
                np.nextafter(2.0 ** -800, 0.0), 2.0 ** 600,
                np.nextafter(2.0 ** 601, 1.0), 2.0 ** 601, 1e300, np.inf,
                -np.inf, np.nan, 1.5e-241, 8.3e180]
    for j, s in enumerate(specials):            # cells 16 ... : one each
        x[16 + j, (37 * j) % K] = s
    fracs = [1.0, 0.5, 0.3, 1.0 / 3.0, 1e-30, 1e-38, 2.0 ** -126,
             np.nextafter(2.0 ** -126, 0.0), 2.0 ** 126,
             np.nextafter(2.0 ** 127, 1.0), 1e38, 1e-310, 0.0, -0.5, 1e300,
             0.9999999999999999]
    row = np.repeat(np.arange(n_cells * len(fracs)), 1)
    col = row % n_cells
    frac_b = np.repeat(np.asarray(fracs), n_cells)
    S = np.ones(row.size)
    n_b = row.size
    plan = engine.RemapPlan.from_triplets(row + 1, col + 1, S, frac_b,
                                          n_cells, n_b, index_base=1,
                                          device=dev)
    csr = oracle.coo_to_csr(row, col, S, n_b, n_cells)
    with np.errstate(all='ignore'):
        for rows, share, tunes in ((None, 0, [None, [1], [6, 2, 2]]),
                                   (4, 0, [[10, 0, 1, 1]]),
                                   (8, 4, [[10, 1, 2, 1, 3],
                                           [10, 0, 2, 0, 3, 32]])):
            if rows:
                plan.build_groups(None, rows=rows, share=share)
            for tune in tunes:
                _check(plan, csr, frac_b, x, dev, engine.MODE_FRACB, 0.0,
                       tune, f'division, rows={rows} tune={tune}')
