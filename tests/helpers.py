"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def golden_files(pattern='g[0125]_*.npz'):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def assert_bitwise(actual, expected, what=''):
    """
    Bit-exact comparison for fp64 results: identical NaN placement, and
    identical bit patterns (sign of zero included) everywhere else.  NaN
    payloads are not compared (x86 and gfx950 pick different default NaNs).
    """
    actual = np.ascontiguousarray(actual, dtype=np.float64)
    expected = np.ascontiguousarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape, \
        f'{what}: shape {actual.shape} != {expected.shape}'
    nan_a = np.isnan(actual)
    nan_e = np.isnan(expected)
    assert np.array_equal(nan_a, nan_e), \
        f'{what}: NaN placement differs at {np.argwhere(nan_a != nan_e)[:5]}'
    bits_a = actual.view(np.int64)[~nan_a]
    bits_e = expected.view(np.int64)[~nan_e]
    bad = bits_a != bits_e
    if bad.any():
        idx = np.flatnonzero(bad)[:5]
        raise AssertionError(
            f'{what}: {bad.sum()} of {bad.size} values differ bitwise, e.g. '
            f'{actual[~nan_a][idx]} vs {expected[~nan_e][idx]}')


def golden_cases(path):
    """Yield (index, field-as-handed-over, remap_axes, thr, out, mask)."""
    g = np.load(path)
    for i in range(int(g['n_cases'])):
        field = g[f'c{i}_field']
        thr = float(g[f'c{i}_thr'])
        thr = None if np.isnan(thr) else thr
        if f'c{i}_in_mask' in g:
            arg = np.ma.masked_array(field, g[f'c{i}_in_mask'])
        elif bool(g[f'c{i}_was_masked_array']):
            arg = np.ma.masked_array(field, np.isnan(field))
        else:
            arg = field
        yield (i, arg, [int(a) for a in g[f'c{i}_remap_axes']], thr,
               g[f'c{i}_out'], g[f'c{i}_mask'])


def golden_map(path):
    g = np.load(path)
    return {k: g[k] for k in ('n_a', 'n_b', 'src_grid_dims', 'dst_grid_dims',
                              'row', 'col', 'S', 'frac_b', 'csr_indptr',
                              'csr_indices', 'csr_data')}
