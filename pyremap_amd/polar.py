"""
Polar stereographic comparison grids, with the names of ``pyremap/polar.py``.
The projections are ``pyproj.Proj`` objects when pyproj is importable (as in
the reference) and this package's own WGS84 polar stereographic otherwise.
"""
import numpy as np

from pyremap_amd.descriptor import ProjectionGridDescriptor
from pyremap_amd.descriptor.projection import (
    antarctic_stereographic,
    arctic_stereographic,
)


def _proj(string, fallback):
    try:
        import pyproj
    except ImportError:
        return fallback()
    return pyproj.Proj(string)


def get_arctic_stereographic_projection():
    """``polar.py:18-36``: standard parallel 75 N."""
    return _proj('+proj=stere +lat_ts=75.0 +lat_0=90 +lon_0=0.0 +k_0=1.0 '
                 '+x_0=0.0 +y_0=0.0 +ellps=WGS84', arctic_stereographic)


def get_antarctic_stereographic_projection():
    """``polar.py:39-49``: standard parallel 71 S."""
    return _proj('+proj=stere +lat_ts=-71.0 +lat_0=-90 +lon_0=0.0 +k_0=1.0 '
                 '+x_0=0.0 +y_0=0.0 +ellps=WGS84', antarctic_stereographic)


def _get_projection(projection):
    """``polar.py:149-159``."""
    if isinstance(projection, str):
        if projection == 'arctic':
            return get_arctic_stereographic_projection()
        if projection == 'antarctic':
            return get_antarctic_stereographic_projection()
        raise ValueError(f'Bad projection name {projection}')
    return projection


def get_polar_descriptor_from_file(filename, projection='antarctic'):
    """
    A descriptor of the x/y grid stored in ``filename``
    (``polar.py:52-83``; the name always says ``antarctic_stereo``, as there).
    """
    from pyremap_amd.io.netcdf import open_dataset
    ds = open_dataset(filename)
    x = np.asarray(ds['x'].values)
    y = np.asarray(ds['y'].values)
    dx = int((x[1] - x[0]) / 1000.0)
    lx = int((x[-1] - x[0]) / 1000.0)
    ly = int((y[-1] - y[0]) / 1000.0)
    mesh_name = f'{lx}x{ly}km_{dx}km_antarctic_stereo'
    return ProjectionGridDescriptor.create(_get_projection(projection), x, y,
                                           mesh_name)


def get_polar_descriptor(lx, ly, dx, dy, projection='antarctic'):
    """
    A polar stereographic grid ``lx`` x ``ly`` km wide with ``dx`` x ``dy`` km
    cells, centred on the pole (``polar.py:86-124``).
    """
    upper = projection[0].upper() + projection[1:]
    mesh_name = f'{lx}x{ly}km_{dx}km_{upper}_stereo'
    x_max = 0.5 * lx * 1e3
    y_max = 0.5 * ly * 1e3
    x = np.linspace(-x_max, x_max, int(lx / dx) + 1)
    y = np.linspace(-y_max, y_max, int(ly / dy) + 1)
    return ProjectionGridDescriptor.create(_get_projection(projection), x, y,
                                           mesh_name)


def _transform(points, projection, to_xy):
    if hasattr(projection, 'inverse'):
        if to_xy:
            a, b = projection.forward(points[:, 0], points[:, 1])
        else:
            a, b = projection.inverse(points[:, 0], points[:, 1])
    else:
        import pyproj
        lat_lon = pyproj.Proj(proj='latlong', datum='WGS84')
        pair = (lat_lon, projection) if to_xy else (projection, lat_lon)
        a, b = pyproj.Transformer.from_proj(*pair).transform(
            points[:, 0], points[:, 1], radians=False)
    points[:, 0] = a
    points[:, 1] = b
    return points


def to_polar(points):
    """(lon, lat) columns -> Antarctic stereographic (x, y), in place
    (``polar.py:127-135``)."""
    return _transform(points, get_antarctic_stereographic_projection(), True)


def from_polar(points):
    """Antarctic stereographic (x, y) columns -> (lon, lat), in place
    (``polar.py:138-146``)."""
    return _transform(points, get_antarctic_stereographic_projection(), False)
