"""
Cell corners of coordinate axes: from CF ``bounds`` variables when a dataset
carries usable ones, otherwise by averaging neighbouring centres and
extrapolating half a cell at the ends.

Behaviour follows ``pyremap/descriptor/utility.py:20-247`` of the reference
(same results, same warnings -- pinned by ``tests/golden/g4_descriptors.npz``);
the implementation is this package's own.
"""
import warnings

import numpy as np


def extrapolate_corners_1d(centres):
    """n centres -> n + 1 corners (``utility.py:220-228``)."""
    c = np.asarray(centres, dtype=np.float64)
    edges = np.empty(c.shape[0] + 1)
    edges[1:-1] = 0.5 * (c[:-1] + c[1:])
    edges[0] = 1.5 * c[0] - 0.5 * c[1]
    edges[-1] = 1.5 * c[-1] - 0.5 * c[-2]
    return edges


def extrapolate_corners_2d(centres):
    """(ny, nx) centres -> (ny + 1, nx + 1) corners: along x first, then
    along y (``utility.py:231-246``; the order matters in the last bits)."""
    c = np.asarray(centres, dtype=np.float64)
    ny, nx = c.shape
    along_x = np.empty((ny, nx + 1))
    along_x[:, 1:-1] = 0.5 * (c[:, :-1] + c[:, 1:])
    along_x[:, 0] = 1.5 * c[:, 0] - 0.5 * c[:, 1]
    along_x[:, -1] = 1.5 * c[:, -1] - 0.5 * c[:, -2]
    out = np.empty((ny + 1, nx + 1))
    out[1:-1, :] = 0.5 * (along_x[:-1, :] + along_x[1:, :])
    out[0, :] = 1.5 * along_x[0, :] - 0.5 * along_x[1, :]
    out[-1, :] = 1.5 * along_x[-1, :] - 0.5 * along_x[-2, :]
    return out


def _cf_bounds(ds, var_name, shape, stacklevel):
    """The variable named by ``ds[var_name].attrs['bounds']`` as a float
    array of ``shape``, or None (with the reference's warnings,
    ``utility.py:103-129``)."""
    name = ds[var_name].attrs.get('bounds')
    if name is None:
        return None
    if name not in ds:
        warnings.warn(
            f'{var_name} has a CF bounds attribute "{name}" but no such '
            f'variable is present in the dataset.', stacklevel=stacklevel)
        return None
    bounds = np.array(ds[name].values, dtype=float)
    if bounds.shape != tuple(shape):
        warnings.warn(
            f'The CF bounds variable {name} has shape {bounds.shape}, not '
            f'the expected {tuple(shape)}.', stacklevel=stacklevel)
        return None
    return bounds


def _tolerance(bounds):
    """
    1e-6 of the largest centre-to-vertex distance (``utility.py:132-136``).
    """
    mid = bounds.mean(axis=-1, keepdims=True)
    return 1e-6 * np.abs(bounds - mid).max()


def _chain_1d(bounds):
    """(n, 2) bounds -> n + 1 corners if consecutive cells touch, in either
    vertex order (``utility.py:139-152``); else None."""
    tol = _tolerance(bounds)
    for b in (bounds, bounds[:, ::-1]):
        if np.all(np.abs(b[:-1, 1] - b[1:, 0]) <= tol):
            return np.append(b[:, 0], b[-1, 1])
    return None


# the 8 ways a cell's 4 vertices can be listed (start vertex x direction), as
# positions of (lower-left, lower-right, upper-right, upper-left) in the list;
# the CF-recommended order comes first so tiny grids fall back to it
_VERTEX_ORDERS = tuple(
    tuple(base[(k + s) % 4] for k in range(4))
    for base in ((0, 1, 2, 3), (0, 3, 2, 1)) for s in range(4))


def _neighbours_agree(b, order, tol):
    ll, lr, ur, ul = order
    pairs = ((b[:, :-1, lr], b[:, 1:, ll]), (b[:, :-1, ur], b[:, 1:, ul]),
             (b[:-1, :, ul], b[1:, :, ll]), (b[:-1, :, ur], b[1:, :, lr]))
    return all(np.all(np.abs(p - q) <= tol) for p, q in pairs)


def _mesh_2d(lat_bounds, lon_bounds):
    """(ny, nx, 4) bounds -> two (ny + 1, nx + 1) corner arrays if some
    vertex order makes neighbours share vertices in BOTH fields
    (``utility.py:155-217``); else None."""
    tol = max(_tolerance(lat_bounds), _tolerance(lon_bounds))
    for order in _VERTEX_ORDERS:
        if not (_neighbours_agree(lat_bounds, order, tol) and
                _neighbours_agree(lon_bounds, order, tol)):
            continue
        ll, lr, ur, ul = order
        out = []
        for b in (lat_bounds, lon_bounds):
            ny, nx = b.shape[:2]
            c = np.zeros((ny + 1, nx + 1))
            c[:-1, :-1] = b[:, :, ll]
            c[:-1, -1] = b[:, -1, lr]
            c[-1, :-1] = b[-1, :, ul]
            c[-1, -1] = b[-1, -1, ur]
            out.append(c)
        return out[0], out[1]
    return None


def corners_1d(ds, var_name):
    """Corners of the 1-D coordinate ``var_name`` (``utility.py:20-53``)."""
    centres = np.array(ds[var_name].values, dtype=float)
    bounds = _cf_bounds(ds, var_name, (len(centres), 2), stacklevel=4)
    if bounds is not None:
        chained = _chain_1d(bounds)
        if chained is not None:
            return chained
        warnings.warn(
            f'The CF bounds of {var_name} are not contiguous so corners '
            f'will be interpolated and extrapolated from cell centers '
            f'instead.', stacklevel=3)
    return extrapolate_corners_1d(centres)


def corners_2d(ds, lat_var_name, lon_var_name):
    """Corner arrays of 2-D lat/lon coordinates (``utility.py:56-100``)."""
    lat = np.array(ds[lat_var_name].values, dtype=float)
    lon = np.array(ds[lon_var_name].values, dtype=float)
    shape = (lat.shape[0], lat.shape[1], 4)
    lat_b = _cf_bounds(ds, lat_var_name, shape, stacklevel=4)
    lon_b = _cf_bounds(ds, lon_var_name, shape, stacklevel=4)
    if lat_b is not None and lon_b is not None:
        mesh = _mesh_2d(lat_b, lon_b)
        if mesh is not None:
            return mesh
        warnings.warn(
            f'The CF bounds of {lat_var_name} and {lon_var_name} do not '
            f'share vertices between neighboring cells so corners will be '
            f'interpolated and extrapolated from cell centers instead.',
            stacklevel=3)
    elif lat_b is not None or lon_b is not None:
        warnings.warn(
            f'Only one of {lat_var_name} and {lon_var_name} has usable CF '
            f'bounds so corners will be interpolated and extrapolated from '
            f'cell centers instead.', stacklevel=3)
    return extrapolate_corners_2d(lat), extrapolate_corners_2d(lon)
