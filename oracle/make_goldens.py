#!/usr/bin/env python3
"""
Generate tests/golden/*.npz from the REFERENCE's own code.

TEST INFRASTRUCTURE ONLY; runs in the build container only (it needs
/root/reference, which does not exist on the GPU box) and is never imported
by the package or by any test.  Re-run with:

    python oracle/make_goldens.py

What is executed is the reference's real code, loaded by file path from
``/root/reference/pyremap/remapper/remap_numpy.py``:

* ``_load_mapping``   (:72-139)  -- through a stub ``xarray.open_dataset``
  that serves the triplets from memory, so the reference's own validation and
  its ``scipy.sparse.csr_matrix((S, (row, col)))`` call build ``_matrix``;
* ``_remap_numpy_array`` (:223-297) -- on seeded fields of several ranks,
  dtypes and modes;
* ``_remap_numpy`` / ``_remap_data_array`` (:19-69, :150-220) -- on
  ``pyremap_amd.xr_lite`` containers installed as the ``xarray`` module the
  reference imports (xarray itself is not installed here).

Nothing of the reference's source is written to the repo: the .npz files hold
inputs and the outputs the reference produced for them.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True  # never write __pycache__ into the reference

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)

from pyremap_amd import xr_lite  # noqa: E402


# --------------------------------------------------------------------------
# load the reference hot-path module against the stub xarray
# --------------------------------------------------------------------------

_MAPS = {}


def _open_dataset(filename):
    """Stub ``xr.open_dataset``: mapping 'files' are served from memory."""
    m = _MAPS[filename]
    ds = xr_lite.Dataset()
    ds['src_grid_dims'] = (('src_grid_rank',), m['src_grid_dims'])
    ds['dst_grid_dims'] = (('dst_grid_rank',), m['dst_grid_dims'])
    ds['col'] = (('n_s',), m['col'])
    ds['row'] = (('n_s',), m['row'])
    ds['S'] = (('n_s',), m['S'])
    ds['frac_b'] = (('n_b',), m['frac_b'])
    # n_a is only a dimension in a real map file; carry it on a dummy var
    ds['area_a'] = (('n_a',), np.zeros(int(m['n_a'])))
    return ds


def load_reference():
    if not os.path.isdir(REF):
        raise SystemExit(f'{REF} not present: goldens can only be generated '
                         f'in the build container')
    stub = types.ModuleType('xarray')
    stub.DataArray = xr_lite.DataArray
    stub.Dataset = xr_lite.Dataset
    stub.open_dataset = _open_dataset
    sys.modules['xarray'] = stub
    path = os.path.join(REF, 'pyremap', 'remapper', 'remap_numpy.py')
    spec = importlib.util.spec_from_file_location('_ref_remap_numpy', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Desc:
    def __init__(self, dims, dim_sizes, coords=None, mesh_name='mesh'):
        self.dims = list(dims)
        self.dim_sizes = list(dim_sizes)
        self.coords = coords if coords is not None else {}
        self.mesh_name = mesh_name


class _Remapper:
    """Duck-typed ``Remapper`` state used by the path (remapper.py:119-137)."""

    def __init__(self, map_filename, src, dst):
        self.map_filename = map_filename
        self.src_descriptor = src
        self.dst_descriptor = dst
        self._ds_map = None
        self._matrix = None


# --------------------------------------------------------------------------
# mapping generators (1-based, unsorted, Fortran-ordered dims: Appendix A)
# --------------------------------------------------------------------------

def make_map(rng, src_dims, dst_dims, nnz_lo, nnz_hi, empty_frac=0.1,
             dup_frac=0.05, signed=False, zero_frac_b=0.05,
             unstable_dups=False):
    """
    src_dims/dst_dims in C order.  Returns the 'file' dict.

    scipy sums duplicate (row, col) entries after a per-row ``std::sort`` by
    column, which is only stable for rows of <= 16 entries.  Unless
    ``unstable_dups`` is set, rows longer than that get at most PAIRS of
    duplicates (a + b is commutative), so the expected sums do not depend on
    the C++ library scipy was built with.
    """
    n_a = int(np.prod(src_dims))
    n_b = int(np.prod(dst_dims))
    rows, cols, vals = [], [], []
    for i in range(n_b):
        if rng.random() < empty_frac:
            continue
        k = int(rng.integers(nnz_lo, nnz_hi + 1))
        centre = int(i * n_a / n_b)
        if unstable_dups:
            cand = (centre + rng.integers(-8, 9, size=k)) % n_a
        else:
            half = max(8, k)
            window = np.arange(-half, half + 1)
            cand = (centre + rng.choice(window, size=k, replace=False)) % n_a
        w = rng.random(k) + 0.05
        if signed:
            w = w * rng.choice([-0.3, 1.0], size=k)
        rows.extend([i] * k)
        cols.extend(cand.tolist())
        vals.extend(w.tolist())
        # explicit duplicates of an existing (row, col)
        n_row = k
        dup_count = {}
        while rng.random() < dup_frac:
            j = int(rng.integers(0, k))
            c = int(cand[j])
            if not unstable_dups:
                if n_row + 1 > 16 and dup_count.get(c, 0) >= 1:
                    continue
                if n_row + 1 > 16 and max(dup_count.values(),
                                          default=0) >= 2:
                    continue
            dup_count[c] = dup_count.get(c, 0) + 1
            n_row += 1
            rows.append(i)
            cols.append(c)
            vals.append(float(rng.random() + 0.01))
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    # row sums -> frac_b in (0, 1]; normalise weights so rows sum to frac_b
    rowsum = np.zeros(n_b)
    np.add.at(rowsum, rows, vals)
    frac_b = np.where(rowsum != 0, rng.random(n_b) * 0.9 + 0.1, 0.0)
    frac_b[rng.random(n_b) < zero_frac_b] = 0.0
    scale = np.where(rowsum != 0, frac_b / np.where(rowsum != 0, rowsum, 1),
                     1.0)
    vals = vals * np.where(scale[rows] != 0, scale[rows], 1.0)
    # shuffle: real files are not row-sorted (multi-PET ESMF output)
    perm = rng.permutation(rows.shape[0])
    return {
        'n_a': np.int64(n_a), 'n_b': np.int64(n_b),
        'src_grid_dims': np.asarray(src_dims[::-1], dtype=np.int32),
        'dst_grid_dims': np.asarray(dst_dims[::-1], dtype=np.int32),
        'row': (rows[perm] + 1).astype(np.int32),
        'col': (cols[perm] + 1).astype(np.int32),
        'S': vals[perm],
        'frac_b': frac_b,
    }


def fill(masked_out):
    """Reference result -> (NaN-filled data, mask, raw data under mask)."""
    mask = np.ma.getmaskarray(masked_out)
    raw = np.array(np.ma.getdata(masked_out), dtype=np.float64, copy=True)
    filled = raw.copy()
    filled[mask] = np.nan
    return np.ascontiguousarray(filled), np.ascontiguousarray(mask), \
        np.ascontiguousarray(raw)


def run_array_cases(ref, name, m, src_names, dst_names, cases):
    """
    cases: list of dicts {field, remap_axes, thr, wrap} where wrap says how
    the field is handed over: 'auto' = as _remap_data_array does (masked
    array iff any NaN), 'plain' = raw ndarray (NaNs propagate), 'ma' = always
    a MaskedArray.
    """
    _MAPS[name] = m
    src_dims = list(m['src_grid_dims'][::-1])
    dst_dims = list(m['dst_grid_dims'][::-1])
    remapper = _Remapper(name, _Desc(src_names, src_dims),
                         _Desc(dst_names, dst_dims))
    ref._load_mapping(remapper)
    csr = remapper._matrix
    out = {k: v for k, v in m.items()}
    out['csr_indptr'] = np.asarray(csr.indptr, dtype=np.int64)
    out['csr_indices'] = np.asarray(csr.indices, dtype=np.int32)
    out['csr_data'] = np.asarray(csr.data, dtype=np.float64)
    out['n_cases'] = np.int64(len(cases))
    for i, c in enumerate(cases):
        field = c['field']
        nanmask = np.isnan(field)
        wrap = c.get('wrap', 'auto')
        if wrap == 'explicit':
            # a caller's own MaskedArray: the mask is NOT isnan(field)
            arg = np.ma.masked_array(field, c['mask'])
            out[f'c{i}_in_mask'] = np.asarray(c['mask'], dtype=np.bool_)
        elif wrap == 'ma' or (wrap == 'auto' and nanmask.any()):
            arg = np.ma.masked_array(field, nanmask)
        else:
            arg = field
        res = ref._remap_numpy_array(remapper, arg, list(c['remap_axes']),
                                     c['thr'])
        filled, mask, raw = fill(res)
        out[f'c{i}_field'] = field
        out[f'c{i}_remap_axes'] = np.asarray(c['remap_axes'], dtype=np.int64)
        out[f'c{i}_thr'] = np.float64(np.nan if c['thr'] is None
                                      else c['thr'])
        out[f'c{i}_wrap'] = np.array(wrap)
        out[f'c{i}_was_masked_array'] = np.bool_(
            isinstance(arg, np.ma.MaskedArray))
        out[f'c{i}_out'] = filled
        out[f'c{i}_mask'] = mask
    path = os.path.join(OUT, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path}: {len(cases)} cases, nnz_in={m["S"].shape[0]} '
          f'nnz_csr={csr.nnz}')


# --------------------------------------------------------------------------
# G5: MaskedArrays whose mask is not isnan(data) (remap_numpy.py:262-266)
# --------------------------------------------------------------------------

def golden_g5(ref):
    rng = np.random.default_rng(55)
    m = make_map(rng, (300,), (12, 15), 1, 6)
    f = rng.standard_normal((300, 6))
    # (a) finite values under the mask; (b) a NaN that is NOT masked (goes
    # through `in_mask * in_field` and poisons every cell it touches);
    # (c) NaNs under the mask; (d) both at once, and a 3-D field
    mask_a = rng.random(f.shape) < 0.25
    f_b = f.copy()
    f_b[17, 2] = np.nan
    f_b[150, 0] = np.nan
    mask_b = rng.random(f.shape) < 0.15
    mask_b[17, 2] = False
    mask_b[150, 0] = False
    f_c = f.copy()
    mask_c = rng.random(f.shape) < 0.3
    f_c[mask_c] = np.nan
    f_d = rng.standard_normal((2, 300, 5))
    mask_d = rng.random(f_d.shape) < 0.2
    f_d[0, 40, :] = np.nan
    mask_d[0, 40, :] = [True, False, True, False, False]
    cases = [
        dict(field=f, mask=mask_a, remap_axes=[0], thr=0.1, wrap='explicit'),
        dict(field=f_b, mask=mask_b, remap_axes=[0], thr=0.1,
             wrap='explicit'),
        dict(field=f_c, mask=mask_c, remap_axes=[0], thr=0.3,
             wrap='explicit'),
        dict(field=f_d, mask=mask_d, remap_axes=[1], thr=0.05,
             wrap='explicit'),
        # no threshold: the mask is ignored altogether (:258-261, :268)
        dict(field=f_b, mask=mask_b, remap_axes=[0], thr=None,
             wrap='explicit'),
    ]
    run_array_cases(ref, 'g5_explicit_mask', m, ['nCells'], ['lat', 'lon'],
                    cases)


# --------------------------------------------------------------------------
# G0: hand-checkable
# --------------------------------------------------------------------------

def golden_g0(ref):
    # n_a = 5, n_b = 4 (2 x 2); row 2 empty; row 3 has frac_b = 0;
    # (row 1, col 2) appears twice; rows unsorted.
    m = {
        'n_a': np.int64(5), 'n_b': np.int64(4),
        'src_grid_dims': np.asarray([5], dtype=np.int32),
        'dst_grid_dims': np.asarray([2, 2], dtype=np.int32),
        'row': np.asarray([4, 1, 2, 1, 2, 4, 2], dtype=np.int32),
        'col': np.asarray([5, 1, 2, 3, 4, 1, 2], dtype=np.int32),
        'S': np.asarray([0.5, 0.25, 0.125, 0.75, 0.5, 0.5, 0.375]),
        'frac_b': np.asarray([1.0, 0.5, 0.0, 0.0]),
    }
    f1 = np.asarray([1.0, 2.0, 4.0, 8.0, 16.0])
    f2 = np.stack([f1, -f1], axis=1)
    f1n = f1.copy()
    f1n[1] = np.nan
    f2n = f2.copy()
    f2n[3, 1] = np.nan
    cases = [
        dict(field=f1, remap_axes=[0], thr=None),
        dict(field=f2, remap_axes=[0], thr=None),
        dict(field=f1n, remap_axes=[0], thr=0.01),
        dict(field=f2n, remap_axes=[0], thr=0.01),
        dict(field=f2n, remap_axes=[0], thr=None),
        dict(field=f2n, remap_axes=[0], thr=0.01, wrap='plain'),
        dict(field=f2, remap_axes=[0], thr=0.3),
    ]
    run_array_cases(ref, 'g0_hand', m, ['n'], ['y', 'x'], cases)


# --------------------------------------------------------------------------
# G1: random, several ranks / axes / dtypes / modes
# --------------------------------------------------------------------------

def with_nans(rng, a, frac):
    a = a.copy()
    a[rng.random(a.shape) < frac] = np.nan
    return a


def golden_g1(ref):
    for seed in range(4):
        rng = np.random.default_rng(seed)
        signed = seed in (1, 3)
        hi = 20 if seed == 3 else 6
        # 1-D source (400) -> 2-D destination (15 x 20)
        m = make_map(rng, [400], [15, 20], 1, hi, signed=signed,
                     dup_frac=0.3 if seed == 3 else 0.05)
        f_1d = rng.standard_normal(400)
        f_first = rng.standard_normal((400, 7))
        f_mid = rng.standard_normal((3, 400, 5))
        f_last = rng.standard_normal((4, 3, 400))
        f_4d = rng.standard_normal((2, 400, 3, 4))
        f_f32 = rng.standard_normal((2, 400, 6)).astype(np.float32)
        f_wide = rng.standard_normal((400, 131))
        cases = [
            dict(field=f_1d, remap_axes=[0], thr=None),
            dict(field=f_first, remap_axes=[0], thr=None),
            dict(field=f_mid, remap_axes=[1], thr=None),
            dict(field=f_last, remap_axes=[2], thr=None),
            dict(field=f_4d, remap_axes=[1], thr=None),
            dict(field=f_f32, remap_axes=[1], thr=None),
            dict(field=f_wide, remap_axes=[0], thr=None),
            dict(field=with_nans(rng, f_1d, 0.2), remap_axes=[0], thr=0.01),
            dict(field=with_nans(rng, f_mid, 0.3), remap_axes=[1], thr=0.01),
            dict(field=with_nans(rng, f_mid, 0.3), remap_axes=[1], thr=0.5),
            dict(field=with_nans(rng, f_last, 0.3), remap_axes=[2],
                 thr=0.01),
            dict(field=with_nans(rng, f_4d, 0.1), remap_axes=[1], thr=0.0),
            dict(field=with_nans(rng, f_f32, 0.3), remap_axes=[1], thr=0.01),
            dict(field=with_nans(rng, f_wide, 0.05), remap_axes=[0],
                 thr=0.01),
            # NaN + thr=None -> unmasked branch on the raw data
            dict(field=with_nans(rng, f_mid, 0.02), remap_axes=[1],
                 thr=None),
            # plain ndarray + threshold -> threshold ignored
            dict(field=with_nans(rng, f_first, 0.02), remap_axes=[0],
                 thr=0.01, wrap='plain'),
            dict(field=f_first, remap_axes=[0], thr=0.01),
            # a MaskedArray without any masked entry + threshold
            dict(field=f_first, remap_axes=[0], thr=0.01, wrap='ma'),
        ]
        run_array_cases(ref, f'g1_seed{seed}_1d_to_2d', m, ['n'],
                        ['y', 'x'], cases)

        # 2-D source (16 x 25) -> 1-D destination (300): two remap axes
        m2 = make_map(rng, [16, 25], [300], 1, hi, signed=signed)
        g_2d = rng.standard_normal((16, 25))
        g_mid = rng.standard_normal((3, 16, 25, 2))
        g_last = rng.standard_normal((5, 16, 25))
        cases2 = [
            dict(field=g_2d, remap_axes=[0, 1], thr=None),
            dict(field=g_mid, remap_axes=[1, 2], thr=None),
            dict(field=g_last, remap_axes=[1, 2], thr=None),
            dict(field=with_nans(rng, g_2d, 0.2), remap_axes=[0, 1],
                 thr=0.01),
            dict(field=with_nans(rng, g_mid, 0.2), remap_axes=[1, 2],
                 thr=0.01),
            dict(field=with_nans(rng, g_last, 0.2), remap_axes=[1, 2],
                 thr=0.2),
        ]
        run_array_cases(ref, f'g1_seed{seed}_2d_to_1d', m2, ['y', 'x'],
                        ['n'], cases2)


def golden_unstable(ref):
    """
    Long rows (> 16 entries) holding 3+ copies of one (row, col): scipy's
    duplicate sums then follow std::sort's unspecified order of equal keys.
    The oracle (stable order) may differ in the last bits; the test for this
    file uses a tolerance and documents the limit.
    """
    rng = np.random.default_rng(99)
    m = make_map(rng, [400], [15, 20], 17, 24, signed=True, dup_frac=0.5,
                 unstable_dups=True)
    cases = [dict(field=rng.standard_normal((400, 3)), remap_axes=[0],
                  thr=None)]
    run_array_cases(ref, 'gx_unstable_dups', m, ['n'], ['y', 'x'], cases)


# --------------------------------------------------------------------------
# G2: QU240-sized with the reference's own fixture fields
# --------------------------------------------------------------------------

def golden_g2(ref):
    from scipy.io import netcdf_file
    rng = np.random.default_rng(240)
    path = os.path.join(REF, 'tests', 'test_interpolate',
                        'timeSeries.0002-01-01.nc')
    with netcdf_file(path, 'r', mmap=False) as nc:
        ssh = np.array(nc.variables['timeMonthly_avg_ssh'][:],
                       dtype=np.float64)
        mld = np.array(nc.variables['timeMonthly_avg_tThreshMLD'][:],
                       dtype=np.float64)
    assert ssh.shape == (1, 7153)
    m = make_map(rng, [7153], [180, 360], 1, 4, empty_frac=0.3,
                 dup_frac=0.01)
    both = np.stack([ssh[0], mld[0]], axis=1)       # (nCells, 2)
    land = both.copy()
    land[rng.random(7153) < 0.25, :] = np.nan
    cases = [
        dict(field=ssh, remap_axes=[1], thr=None),
        dict(field=mld, remap_axes=[1], thr=0.01),
        dict(field=land, remap_axes=[0], thr=0.01),
    ]
    run_array_cases(ref, 'g2_qu240_to_1deg', m, ['nCells'], ['lat', 'lon'],
                    cases)


# --------------------------------------------------------------------------
# G6: the REAL QU240 cell numbering (icosahedral-bisection order, nothing
# like the destination raster): cell centres from the reference's mesh
# fixture, overlap-like weights to the 1-degree grid by nearest centres,
# columns = the mesh's own cell ids, the reference's fixture fields
# --------------------------------------------------------------------------

def golden_g6(ref):
    import torch
    from scipy.io import netcdf_file

    from pyremap_amd import synthetic
    base = os.path.join(REF, 'tests', 'test_interpolate')
    with netcdf_file(os.path.join(base, 'mpasMesh.nc'), 'r',
                     mmap=False) as nc:
        lat = np.array(nc.variables['latCell'][:], dtype=np.float64)
        lon = np.array(nc.variables['lonCell'][:], dtype=np.float64)
    cells = os.path.join(OUT, 'qu240_cells.npz')
    np.savez_compressed(cells, latCell=lat, lonCell=lon)
    print(f'wrote {cells}: {lat.shape[0]} cell centres')
    with netcdf_file(os.path.join(base, 'timeSeries.0002-01-01.nc'), 'r',
                     mmap=False) as nc:
        ssh = np.array(nc.variables['timeMonthly_avg_ssh'][:],
                       dtype=np.float64)
        mld = np.array(nc.variables['timeMonthly_avg_tThreshMLD'][:],
                       dtype=np.float64)
    sm = synthetic.knn_map(torch.from_numpy(lat), torch.from_numpy(lon),
                           (180, 360), k_hi=4, seed=6)
    m = sm.numpy()
    rng = np.random.default_rng(606)
    both = np.stack([ssh[0], mld[0]], axis=1)       # (nCells, 2)
    land = both.copy()
    land[rng.random(7153) < 0.25, :] = np.nan
    cases = [
        dict(field=ssh, remap_axes=[1], thr=None),
        dict(field=mld, remap_axes=[1], thr=0.01),
        dict(field=land, remap_axes=[0], thr=0.01),
    ]
    run_array_cases(ref, 'g6_qu240_real_numbering', m, ['nCells'],
                    ['lat', 'lon'], cases)


# --------------------------------------------------------------------------
# G3: Dataset / DataArray level (a2, a3, a4) and the error messages
# --------------------------------------------------------------------------

def ds_to_record(ds):
    rec = {'attrs': {k: str(v) for k, v in ds.attrs.items()},
           'data_vars': [], 'coords': []}
    arrays = {}
    for kind, names in (('data_vars', list(ds.data_vars)),
                        ('coords', list(ds.coords))):
        for nm in names:
            var = ds.variables[nm]
            rec[kind].append({'name': nm, 'dims': list(var.dims),
                              'attrs': {k: str(v)
                                        for k, v in var.attrs.items()},
                              'dtype': str(var.dtype)})
            arrays[nm] = var.values
    return rec, arrays


def golden_g3(ref):
    rng = np.random.default_rng(33)
    nlat, nlon = 6, 8
    lat = np.linspace(-75.0, 75.0, nlat)
    lon = np.linspace(-157.5, 157.5, nlon)
    dst_coords = {
        'lat': {'dims': 'lat', 'data': lat,
                'attrs': {'units': 'degrees_north'}},
        'lon': {'dims': 'lon', 'data': lon,
                'attrs': {'units': 'degrees_east'}},
    }
    n_cells = 50
    m = make_map(rng, [n_cells], [nlat, nlon], 1, 4)
    name = 'g3_dataset'
    _MAPS[name] = m

    def new_remapper():
        return _Remapper(name, _Desc(['nCells'], [n_cells]),
                         _Desc(['lat', 'lon'], [nlat, nlon], dst_coords,
                               mesh_name='toy_6x8'))

    ds = xr_lite.Dataset(attrs={'history': 'created by a test',
                                'title': 'toy'})
    temp = rng.standard_normal((3, n_cells, 4))
    temp[:, rng.random(n_cells) < 0.3, :] = np.nan
    ds['temperature'] = xr_lite.DataArray(
        temp, dims=('Time', 'nCells', 'nVertLevels'),
        attrs={'units': 'C', 'long_name': 'temperature'})
    ds['ssh'] = xr_lite.DataArray(
        rng.standard_normal((3, n_cells)), dims=('Time', 'nCells'),
        attrs={'units': 'm'})
    ds['area'] = xr_lite.DataArray(
        rng.random(n_cells).astype(np.float32), dims=('nCells',))
    ds['daysSinceStart'] = xr_lite.DataArray(
        np.arange(3, dtype=np.float64), dims=('Time',))
    ds['refZ'] = xr_lite.DataArray(
        -np.arange(4, dtype=np.float64), dims=('nVertLevels',))
    ds['counter'] = xr_lite.DataArray(
        np.arange(3, dtype=np.int32), dims=('Time',))
    ds._set_coord('Time', xr_lite.DataArray(
        np.asarray([10.0, 20.0, 30.0]), dims=('Time',)))
    ds._set_coord('latCell', xr_lite.DataArray(
        rng.random(n_cells), dims=('nCells',)))

    out = {k: v for k, v in m.items()}
    meta = {}
    in_rec, in_arr = ds_to_record(ds)
    meta['input'] = in_rec
    for k, v in in_arr.items():
        out[f'in__{k}'] = v

    old_argv = sys.argv
    sys.argv = ['golden_prog', '--flag']
    try:
        for tag, thr in (('thr', 0.01), ('nothr', None)):
            res = ref._remap_numpy(new_remapper(), ds, thr)
            rec, arr = ds_to_record(res)
            meta[f'dataset_{tag}'] = rec
            for k, v in arr.items():
                out[f'{tag}__{k}'] = v
        # DataArray in, DataArray out
        res = ref._remap_numpy(new_remapper(), ds['temperature'], 0.01)
        meta['dataarray_thr'] = {
            'name': res.name, 'dims': list(res.dims),
            'attrs': {k: str(v) for k, v in res.attrs.items()},
            'coords': [{'name': k, 'dims': list(v.dims)}
                       for k, v in res.coords.items()]}
        out['da__data'] = res.values
        for k, v in res.coords.items():
            out[f'da_coord__{k}'] = v.values
    finally:
        sys.argv = old_argv

    # error behaviour (messages recorded verbatim)
    errors = {}

    def record(tag, fn):
        try:
            fn()
            errors[tag] = None
        except Exception as exc:  # noqa: BLE001
            errors[tag] = {'type': type(exc).__name__, 'message': str(exc)}

    r = new_remapper()
    r.map_filename = None
    record('no_map', lambda: ref._remap_numpy(r, ds, None))
    r = _Remapper(name, _Desc(['y', 'x'], [5, 10]),
                  _Desc(['lat', 'lon'], [nlat, nlon], dst_coords))
    record('src_rank', lambda: ref._remap_numpy(r, ds, None))
    r = _Remapper(name, _Desc(['nCells'], [n_cells]),
                  _Desc(['n'], [nlat * nlon], {}))
    record('dst_rank', lambda: ref._remap_numpy(r, ds, None))
    r = _Remapper(name, _Desc(['nCells'], [n_cells + 1]),
                  _Desc(['lat', 'lon'], [nlat, nlon], dst_coords))
    record('src_size', lambda: ref._remap_numpy(r, ds, None))
    r = _Remapper(name, _Desc(['nCells'], [n_cells]),
                  _Desc(['lat', 'lon'], [nlon, nlat], dst_coords))
    record('dst_size', lambda: ref._remap_numpy(r, ds, None))
    ds_bad = xr_lite.Dataset()
    ds_bad['ssh'] = xr_lite.DataArray(np.zeros((3, n_cells + 2)),
                                      dims=('Time', 'nCells'))
    record('ds_size', lambda: ref._remap_numpy(new_remapper(), ds_bad, None))
    record('not_xarray',
           lambda: ref._remap_numpy(new_remapper(),
                                    {'sizes': 1}, None))

    class _Sized:
        sizes = {'nCells': n_cells}
    record('type_error', lambda: ref._remap_numpy(new_remapper(), _Sized(),
                                                  None))
    # partial source dims (2-D source, array holding only one of them)
    m2 = make_map(rng, [5, 10], [nlat, nlon], 1, 3)
    _MAPS['g3_partial'] = m2
    r2 = _Remapper('g3_partial', _Desc(['y', 'x'], [5, 10]),
                   _Desc(['lat', 'lon'], [nlat, nlon], dst_coords,
                         mesh_name='toy_6x8'))
    ds2 = xr_lite.Dataset()
    ds2['full'] = xr_lite.DataArray(rng.standard_normal((2, 5, 10)),
                                    dims=('t', 'y', 'x'))
    ds2['only_y'] = xr_lite.DataArray(rng.standard_normal((5,)),
                                      dims=('y',))
    ds2['only_x'] = xr_lite.DataArray(rng.standard_normal((2, 10)),
                                      dims=('t', 'x'))
    ds2['scalar_t'] = xr_lite.DataArray(rng.standard_normal((2,)),
                                        dims=('t',))
    res2 = ref._remap_numpy(r2, ds2, None)
    rec2, arr2 = ds_to_record(res2)
    meta['partial_dataset'] = rec2
    for k, v in m2.items():
        out[f'p_map__{k}'] = v
    rec2in, arr2in = ds_to_record(ds2)
    meta['partial_input'] = rec2in
    for k, v in arr2in.items():
        out[f'p_in__{k}'] = v
    for k, v in arr2.items():
        out[f'p_out__{k}'] = v
    record('partial_dataarray',
           lambda: ref._remap_data_array(ds2['only_y'], r2, None))
    meta['errors'] = errors
    meta['argv'] = ['golden_prog', '--flag']
    meta['dst_coords'] = {k: {'dims': v['dims'], 'attrs': v['attrs']}
                          for k, v in dst_coords.items()}
    out['dst_lat'] = lat
    out['dst_lon'] = lon
    out['meta_json'] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path}')
    for k, v in errors.items():
        print('   error', k, '->', v)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    golden_g0(ref)
    golden_g1(ref)
    golden_g2(ref)
    golden_g3(ref)
    golden_unstable(ref)
    golden_g5(ref)
    golden_g6(ref)


if __name__ == '__main__':
    main()
