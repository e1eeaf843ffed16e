"""
GPU parity on REALISTIC source-cell numberings:
col is whatever the mesh says (remap_numpy.py:134-137), and no MPAS mesh is
numbered along the destination raster.  Bitwise against the oracle:

* the real QU240 numbering (tests/golden/g6_qu240_real_numbering.npz: cell
  centres and ids of the reference's mesh fixture -> 1 degree) as BASELINE
  config 2 runs, K = 64, every row;
* config 3's overlaps with the source cells numbered as an MPAS mesh numbers
  them (synthetic.mesh_numbering) and at random, every row at K = 128;
  the schedule is whatever `remap_schedule_auto` picks.
"""
import os

import numpy as np
import pytest

from helpers import assert_bitwise, golden_map

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from pyremap_amd import engine
    engine.load_library()
    return torch.device('cuda', 0)


@pytest.mark.parametrize('mode', ['fracb', 'masked'])
def test_config2_with_the_real_qu240_numbering(dev, golden_dir, mode):
    from oracle import oracle
    from pyremap_amd import engine
    m = golden_map(os.path.join(golden_dir, 'g6_qu240_real_numbering.npz'))
    n_a, n_b = int(m['n_a']), int(m['n_b'])
    dst_dims = tuple(int(d) for d in m['dst_grid_dims'][::-1])
    assert (n_a, dst_dims) == (7153, (180, 360))
    plan = engine.RemapPlan.from_triplets(m['row'], m['col'], m['S'],
                                          m['frac_b'], n_a, n_b, device=dev)
    rowptr, col, val = plan.to_host_csr()
    assert np.array_equal(rowptr, m['csr_indptr'])
    assert np.array_equal(col, m['csr_indices'])
    assert np.array_equal(val, m['csr_data'])
    # ids met by one destination latitude row: most of the range
    r0, r1 = rowptr[90 * 360], rowptr[91 * 360]
    assert col[r0:r1].max() - col[r0:r1].min() > 0.6 * n_a
    choice = plan.auto_schedule(dst_dims)
    csr = oracle.OracleCSR(rowptr, col, val, (n_b, n_a))
    rng = np.random.default_rng(66)
    masked = mode == 'masked'
    emode = engine.MODE_MASKED if masked else engine.MODE_FRACB
    for shape, axes in (((n_a, 64), [0]), ((4, n_a, 16), [1]),
                        ((2, n_a, 61), [1]), ((12, n_a), [1])):
        x = rng.standard_normal(shape)
        if masked:
            dead = rng.random(n_a) < 0.2
            x[(slice(None),) * axes[0] + (dead,)] = np.nan
        arg = np.ma.masked_array(x, np.isnan(x)) if masked else x
        ref = oracle.remap_numpy_array(csr, m['frac_b'], dst_dims, arg, axes,
                                       0.01 if masked else None)
        y = engine.remap_tensor(plan, dst_dims, torch.from_numpy(x).to(dev),
                                axes, emode, threshold=0.01)
        assert_bitwise(y.cpu().numpy(), np.ma.filled(ref, np.nan),
                       f'real QU240 numbering {mode} {shape} '
                       f'{choice["family"]}')


@pytest.mark.parametrize('locality', ['mesh', 'scatter'])
def test_config3_renumbered_every_row_bitwise(dev, locality):
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    m = synthetic.make_config('config3', device=dev, locality=locality)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    choice = plan.auto_schedule(m.dst_dims)
    # the geometry is untouched by the renumbering: rows that are neighbours
    # still share source cells, so the row groups are still what is chosen
    assert choice['family'] == 'rowgroup', choice
    K = 128
    g = torch.Generator(device=dev)
    g.manual_seed(33)
    x = torch.randn((m.n_a, K), generator=g, device=dev, dtype=torch.float64)
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    for masked in (False, True):
        if masked:
            x[torch.rand(m.n_a, generator=g, device=dev) < 0.2, :] = \
                float('nan')
        y = engine.remap_tensor(
            plan, m.dst_dims, x, [0],
            engine.MODE_MASKED if masked else engine.MODE_FRACB,
            threshold=0.01)
        ref, ref_mask = oracle.remap_flat(csr, frac_b, x.cpu().numpy(),
                                          masked, 0.01,
                                          nthreads=os.cpu_count() or 1)
        ref[ref_mask] = np.nan
        assert_bitwise(y.cpu().numpy().reshape(m.n_b, K), ref,
                       f'config3 {locality} masked={masked}')
