"""
Pins the CPU oracle (oracle/) against golden vectors produced by the
reference's own `_load_mapping` and `_remap_numpy_array`
(pyremap/remapper/remap_numpy.py:72-139, 223-297; see oracle/make_goldens.py).
Bit-exact: fp64 values identical, NaN / mask placement identical.
"""
import os

import numpy as np
import pytest

from helpers import assert_bitwise, golden_cases, golden_files, golden_map
from oracle import oracle

FILES = golden_files()


def test_goldens_present():
    assert len(FILES) >= 10


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)
                                             for f in FILES])
def test_coo_to_csr_matches_scipy(path):
    m = golden_map(path)
    csr = oracle.coo_to_csr(m['row'] - 1, m['col'] - 1, m['S'],
                            int(m['n_b']), int(m['n_a']))
    assert np.array_equal(csr.indptr, m['csr_indptr'])
    assert np.array_equal(csr.indices, m['csr_indices'])
    assert_bitwise(csr.data, m['csr_data'], 'csr data')


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)
                                             for f in FILES])
@pytest.mark.parametrize('nthreads', [1, 3])
def test_remap_numpy_array_matches_reference(path, nthreads):
    m = golden_map(path)
    csr = oracle.coo_to_csr(m['row'] - 1, m['col'] - 1, m['S'],
                            int(m['n_b']), int(m['n_a']))
    dst_dims = m['dst_grid_dims'][::-1]
    n = 0
    for i, arg, axes, thr, out, mask in golden_cases(path):
        res = oracle.remap_numpy_array(csr, m['frac_b'], dst_dims, arg, axes,
                                       thr, nthreads=nthreads)
        assert isinstance(res, np.ma.MaskedArray)
        assert res.shape == out.shape, f'case {i}'
        assert np.array_equal(np.ma.getmaskarray(res), mask), f'case {i}'
        assert_bitwise(res.filled(np.nan), out, f'{path} case {i}')
        n += 1
    assert n > 0


def test_csr_matvecs_is_sequential_mul_add():
    rng = np.random.default_rng(5)
    m = golden_map(FILES[1])
    csr = oracle.coo_to_csr(m['row'] - 1, m['col'] - 1, m['S'],
                            int(m['n_b']), int(m['n_a']))
    X = rng.standard_normal((int(m['n_a']), 3))
    Y = oracle.csr_matvecs(csr, X)
    ref = np.zeros_like(Y)
    for i in range(csr.shape[0]):
        for jj in range(csr.indptr[i], csr.indptr[i + 1]):
            ref[i] = ref[i] + csr.data[jj] * X[csr.indices[jj]]
    assert_bitwise(Y, ref)


def test_unstable_duplicate_order_is_a_documented_limit(golden_dir):
    """
    Rows with > 16 entries AND >= 3 copies of one (row, col): the reference's
    sums follow std::sort's unspecified order of equal keys (scipy
    csr_sort_indices), so only closeness is asserted (weights are O(1) and
    signed, hence an absolute bound of a few ulp of 1.0).
    """
    path = os.path.join(golden_dir, 'gx_unstable_dups.npz')
    m = golden_map(path)
    csr = oracle.coo_to_csr(m['row'] - 1, m['col'] - 1, m['S'],
                            int(m['n_b']), int(m['n_a']))
    assert np.array_equal(csr.indptr, m['csr_indptr'])
    assert np.array_equal(csr.indices, m['csr_indices'])
    np.testing.assert_allclose(csr.data, m['csr_data'], rtol=1e-13,
                               atol=2e-15)
    assert not np.array_equal(csr.data, m['csr_data']), \
        'the fixture no longer exercises the unstable-order case'
