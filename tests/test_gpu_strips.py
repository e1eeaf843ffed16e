"""
Kernel family 8 (csrc/spmm_strip.h): an LDS ring of source-row pieces sliding
along strips of the destination grid, loader waves + compute waves, for
entry-rich mappings (BASELINE config 5's kind: 2nd-order conservative
stencils).  Every value against the oracle, bit for bit, through the C ABI.
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


def _problem(dev, n_a=1200, dims=(37, 61), k=(6, 20), seed=3,
             locality='mesh'):
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(n_a, dims, k[0], k[1], seed=seed,
                                   signed=True, locality=locality)
    mm = m.numpy()
    plan = engine.RemapPlan.from_triplets(
        mm['row'], mm['col'], mm['S'], mm['frac_b'], m.n_a, m.n_b,
        index_base=1, device=dev)
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    return m, mm, plan, csr


@pytest.mark.parametrize('shape', [
    dict(strip_rows=8, step_cols=2, segments=3, depth=2, waves=8),
    dict(strip_rows=14, step_cols=1, segments=1, depth=1, waves=14),
    dict(strip_rows=4, step_cols=4, segments=2, depth=3, waves=13, gap=0),
])
@pytest.mark.parametrize('K', [64, 100, 512])
def test_strip_kernel_bitwise(dev, shape, K):
    from oracle import oracle
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev)
    plan.build_strips(m.dst_dims, **shape)
    rng = np.random.default_rng(K)
    x = rng.standard_normal((m.n_a, K))
    x[rng.random(m.n_a) < 0.15] = np.nan
    xd = torch.from_numpy(x).to(dev)
    for mode, masked, thr in ((engine.MODE_FRACB, False, 0.0),
                              (engine.MODE_MASKED, True, 0.01),
                              (engine.MODE_RAW, False, 0.0)):
        y, mask = engine.remap_tensor(plan, None, xd, [0], mode,
                                      threshold=thr, tune=[8],
                                      want_mask=True)
        if mode == engine.MODE_RAW:
            # the bare product (remap_numpy.py:268): NaNs propagate
            assert_bitwise(y.cpu().numpy(), oracle.csr_matvecs(csr, x),
                           'raw')
            continue
        ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, masked, thr)
        ref = ref.copy()
        ref[ref_mask] = np.nan
        assert_bitwise(y.cpu().numpy(), ref, f'mode {mode}')
        assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask)


def test_strip_kernel_fma_and_fallback(dev):
    """REMAP_FLAG_FMA on the strip family (rtol 1e-13, not the bits); calls
    the family cannot serve -- several batches, float32 -- take the plan's
    other schedule when the strips are a hint, and are refused when they
    are demanded."""
    from oracle import oracle
    from pyremap_amd import engine
    m, mm, plan, csr = _problem(dev, seed=8)
    plan.auto_schedule(m.dst_dims)
    plan.build_strips(m.dst_dims, strip_rows=8, step_cols=2, segments=2,
                      waves=8)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((m.n_a, 128))
    xd = torch.from_numpy(x).to(dev)
    ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], x, False, 0.0)
    y = engine.remap_tensor(plan, None, xd, [0], engine.MODE_FRACB,
                            tune=[8], flags=engine.FLAG_FMA).cpu().numpy()
    ok = ~ref_mask
    assert np.array_equal(np.isnan(y), ref_mask)
    scale = np.abs(ref[ok]).max()
    assert np.abs(y[ok] - ref[ok]).max() <= 1e-13 * scale
    # (Time, nCells, levels): several batches
    x3 = rng.standard_normal((3, m.n_a, 64))
    x3d = torch.from_numpy(x3).to(dev)
    with pytest.raises(engine.EngineError):
        engine.remap_tensor(plan, m.dst_dims, x3d, [1], engine.MODE_FRACB,
                            tune=[8])
    y3 = engine.remap_tensor(plan, m.dst_dims, x3d, [1], engine.MODE_FRACB,
                             tune=[8], flags=engine.FLAG_TUNE_HINT)
    ref3 = np.ma.filled(oracle.remap_numpy_array(
        csr, mm['frac_b'], m.dst_dims, x3, [1], None), np.nan)
    assert_bitwise(y3.cpu().numpy(), ref3, '(T, n, L) through the hint')
    y32 = engine.remap_tensor(plan, None, xd.to(torch.float32), [0],
                              engine.MODE_FRACB, tune=[8],
                              flags=engine.FLAG_TUNE_HINT)
    ref32, m32 = oracle.remap_flat(csr, mm['frac_b'],
                                   x.astype(np.float32).astype(np.float64),
                                   False, 0.0)
    ref32 = ref32.copy()
    ref32[m32] = np.nan
    assert_bitwise(y32.cpu().numpy(), ref32, 'float32 through the hint')
