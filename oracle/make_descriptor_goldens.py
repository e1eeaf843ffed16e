#!/usr/bin/env python3
"""
Golden vectors for the descriptor constructors,
captured from the REFERENCE's own classes run in this container.

TEST INFRASTRUCTURE ONLY.  Needs /root/reference (absent on the GPU box); the
output ``tests/golden/g4_descriptors.npz`` is what travels.

The reference's descriptor modules import xarray, pyproj and netCDF4, none of
which is installed here.  They are loaded file by file under their real module
names against stubs: ``xarray`` = this package's ``xr_lite`` containers,
``pyproj`` / ``pyremap.utility`` = empty shells (nothing on the exercised paths
-- ``LatLonGridDescriptor.read/create``, ``LatLon2DGridDescriptor.read``,
``get_lat_lon_descriptor``, ``utility.get_corners_*`` -- touches them).

Each case stores the dataset it was read from (variables, dims, attrs) and the
descriptor the reference produced: lat, lon, corners, and a JSON record of
``mesh_name, regional, dims, dim_sizes, units`` and the warnings raised.

    python oracle/make_descriptor_goldens.py
"""
import importlib.util
import json
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden', 'g4_descriptors.npz')
sys.path.insert(0, REPO)

from pyremap_amd import xr_lite  # noqa: E402


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    if not os.path.isdir(REF):
        raise SystemExit(f'{REF} not present: goldens can only be generated '
                         f'in the build container')
    xr = types.ModuleType('xarray')
    xr.DataArray = xr_lite.DataArray
    xr.Dataset = xr_lite.Dataset
    sys.modules['xarray'] = xr
    pyproj = types.ModuleType('pyproj')
    pyproj.__path__ = []
    pyproj.Transformer = object
    pyproj.enums = types.ModuleType('pyproj.enums')
    sys.modules['pyproj'] = pyproj
    sys.modules['pyproj.enums'] = pyproj.enums
    for pkg in ('pyremap', 'pyremap.descriptor'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    util = types.ModuleType('pyremap.utility')
    util.write_netcdf = None
    sys.modules['pyremap.utility'] = util
    _load('pyremap.descriptor.utility', 'pyremap/descriptor/utility.py')
    _load('pyremap.descriptor.mesh_descriptor',
          'pyremap/descriptor/mesh_descriptor.py')
    one = _load('pyremap.descriptor.lat_lon_grid_descriptor',
                'pyremap/descriptor/lat_lon_grid_descriptor.py')
    two = _load('pyremap.descriptor.lat_lon_2d_grid_descriptor',
                'pyremap/descriptor/lat_lon_2d_grid_descriptor.py')
    return one, two


# --------------------------------------------------------------------------
# cases
# --------------------------------------------------------------------------

def _bounds_1d(edges, flip=False):
    b = np.stack([edges[:-1], edges[1:]], axis=1)
    return b[:, ::-1].copy() if flip else b


def cases_1d(rng):
    out = []
    # plain global 2-degree grid in degrees, no bounds
    lat = np.linspace(-89.0, 89.0, 90)
    lon = np.linspace(-179.0, 179.0, 180)
    out.append(dict(name='plain_global', lat=lat, lon=lon,
                    units='degrees_north'))
    # non-uniform latitudes with contiguous bounds (not the midpoints)
    edges = np.sort(rng.uniform(-80, 80, 25))
    lat = 0.3 * edges[:-1] + 0.7 * edges[1:]
    lon_e = np.linspace(0.0, 360.0, 37)
    lon = 0.5 * (lon_e[:-1] + lon_e[1:])
    out.append(dict(name='bounds_contiguous', lat=lat, lon=lon,
                    units='degrees', lat_bnds=_bounds_1d(edges),
                    lon_bnds=_bounds_1d(lon_e)))
    # descending latitude, bounds listed upper -> lower
    out.append(dict(name='bounds_descending_flipped', lat=lat[::-1].copy(),
                    lon=lon, units='degrees',
                    lat_bnds=_bounds_1d(edges)[::-1].copy(),
                    lon_bnds=_bounds_1d(lon_e, flip=True)))
    # gaps between cells: warning + extrapolation
    gap = _bounds_1d(edges).copy()
    gap[:, 1] -= 0.2
    out.append(dict(name='bounds_with_gaps', lat=lat, lon=lon,
                    units='degrees', lat_bnds=gap))
    # regional grid in radians
    lat = np.radians(np.linspace(10.0, 40.0, 16))
    lon = np.radians(np.linspace(100.0, 160.0, 31))
    out.append(dict(name='regional_radians', lat=lat, lon=lon,
                    units='radians'))
    # bounds attribute naming a variable that is not there
    out.append(dict(name='bounds_missing', lat=np.linspace(-60, 60, 13),
                    lon=np.linspace(-180.0, 170.0, 36), units='degrees',
                    lat_bounds_attr='nothing_here'))
    # bounds of the wrong shape
    out.append(dict(name='bounds_wrong_shape', lat=np.linspace(-60, 60, 13),
                    lon=np.linspace(-180.0, 170.0, 36), units='degrees',
                    lat_bnds=np.zeros((12, 2))))
    # duplicate end point (-180 and 180 both present), name from the file
    out.append(dict(name='duplicate_endpoint_named',
                    lat=np.linspace(-90, 90, 19),
                    lon=np.linspace(-180.0, 180.0, 37), units='degrees',
                    attrs={'meshName': 'my_grid', 'history': 'made by hand'}))
    # caller-supplied name and regional flag, other variable / dim names
    out.append(dict(name='explicit', lat=np.linspace(-30, 30, 7),
                    lon=np.linspace(0.0, 90.0, 10), units='degrees_east',
                    mesh_name='given', regional=False, lat_var='latitude',
                    lon_var='longitude', lat_dim='nlat', lon_dim='nlon'))
    return out


def _vertex_field(corner, order):
    """(ny+1, nx+1) corners -> (ny, nx, 4) bounds with the cell's
    (ll, lr, ur, ul) vertices stored at positions order[0..3]."""
    ny, nx = corner.shape[0] - 1, corner.shape[1] - 1
    b = np.zeros((ny, nx, 4))
    ll, lr, ur, ul = order
    b[:, :, ll] = corner[:-1, :-1]
    b[:, :, lr] = corner[:-1, 1:]
    b[:, :, ur] = corner[1:, 1:]
    b[:, :, ul] = corner[1:, :-1]
    return b


def cases_2d(rng):
    out = []
    ny, nx = 7, 9
    jj, ii = np.meshgrid(np.arange(ny + 1.0), np.arange(nx + 1.0),
                         indexing='ij')
    # a gently rotated, stretched mesh of corners
    lat_c = 40.0 + 1.1 * jj + 0.15 * ii + 0.02 * jj * ii
    lon_c = -20.0 + 1.7 * ii - 0.2 * jj + 0.01 * ii * ii
    lat = 0.25 * (lat_c[:-1, :-1] + lat_c[:-1, 1:] + lat_c[1:, 1:] +
                  lat_c[1:, :-1])
    lon = 0.25 * (lon_c[:-1, :-1] + lon_c[:-1, 1:] + lon_c[1:, 1:] +
                  lon_c[1:, :-1])
    out.append(dict(name='2d_plain', lat=lat, lon=lon, units='degrees'))
    out.append(dict(name='2d_bounds_cf_order', lat=lat, lon=lon,
                    units='degrees',
                    lat_bnds=_vertex_field(lat_c, (0, 1, 2, 3)),
                    lon_bnds=_vertex_field(lon_c, (0, 1, 2, 3))))
    out.append(dict(name='2d_bounds_clockwise_from_ur', lat=lat, lon=lon,
                    units='degrees',
                    lat_bnds=_vertex_field(lat_c, (3, 2, 1, 0)),
                    lon_bnds=_vertex_field(lon_c, (3, 2, 1, 0))))
    shaken = _vertex_field(lat_c, (0, 1, 2, 3)) + \
        0.05 * rng.standard_normal((ny, nx, 4))
    out.append(dict(name='2d_bounds_not_shared', lat=lat, lon=lon,
                    units='degrees', lat_bnds=shaken,
                    lon_bnds=_vertex_field(lon_c, (0, 1, 2, 3))))
    out.append(dict(name='2d_bounds_only_lat', lat=lat, lon=lon,
                    units='degrees',
                    lat_bnds=_vertex_field(lat_c, (0, 1, 2, 3))))
    out.append(dict(name='2d_radians_global_flag', lat=np.radians(lat),
                    lon=np.radians(lon), units='radians', regional=False,
                    mesh_name='curvy', lat_dim='j', lon_dim='i'))
    return out


def build_dataset(c, two_d):
    ds = xr_lite.Dataset()
    lat_var = c.get('lat_var', 'lat')
    lon_var = c.get('lon_var', 'lon')
    lat_dim = c.get('lat_dim', 'lat' if not two_d else 'y')
    lon_dim = c.get('lon_dim', 'lon' if not two_d else 'x')
    if two_d:
        ds[lat_var] = ((lat_dim, lon_dim), c['lat'])
        ds[lon_var] = ((lat_dim, lon_dim), c['lon'])
    else:
        ds[lat_var] = ((lat_dim,), c['lat'])
        ds[lon_var] = ((lon_dim,), c['lon'])
    ds[lat_var].attrs['units'] = c['units']
    ds[lon_var].attrs['units'] = c['units']
    for which, var, dims in (('lat', lat_var, (lat_dim,)),
                             ('lon', lon_var, (lon_dim,))):
        if two_d:
            dims = (lat_dim, lon_dim)
        key = f'{which}_bnds'
        if key in c:
            b = np.asarray(c[key])
            fits = b.shape[:-1] == np.asarray(c[which]).shape
            bdims = dims if fits else tuple(f'{d}_odd' for d in dims)
            ds[key] = (bdims + ('nv',), b)
            ds[var].attrs['bounds'] = key
        if f'{which}_bounds_attr' in c:
            ds[var].attrs['bounds'] = c[f'{which}_bounds_attr']
    for k, v in c.get('attrs', {}).items():
        ds.attrs[k] = v
    return ds, lat_var, lon_var


def main():
    one, two = load_reference()
    rng = np.random.default_rng(20260804)
    store = {}
    index = []
    sys.argv = ['golden']      # the history attribute records argv
    for two_d, cases in ((False, cases_1d(rng)), (True, cases_2d(rng))):
        cls = two.LatLon2DGridDescriptor if two_d else \
            one.LatLonGridDescriptor
        for c in cases:
            ds, lat_var, lon_var = build_dataset(c, two_d)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter('always')
                d = cls.read(ds=ds, lat_var_name=lat_var,
                             lon_var_name=lon_var,
                             mesh_name=c.get('mesh_name'),
                             regional=c.get('regional'))
            name = c['name']
            for k in ('lat', 'lon', 'lat_bnds', 'lon_bnds'):
                if k in c:
                    store[f'{name}/in/{k}'] = np.asarray(c[k], dtype=float)
            for k in ('lat', 'lon', 'lat_corner', 'lon_corner'):
                store[f'{name}/out/{k}'] = np.asarray(getattr(d, k))
            meta = dict(
                two_d=two_d, units_attr=c['units'], lat_var=lat_var,
                lon_var=lon_var, lat_dim=ds[lat_var].dims[0],
                lon_dim=ds[lon_var].dims[-1],
                lat_bounds_attr=ds[lat_var].attrs.get('bounds'),
                lon_bounds_attr=ds[lon_var].attrs.get('bounds'),
                ds_attrs=c.get('attrs', {}),
                arg_mesh_name=c.get('mesh_name'),
                arg_regional=c.get('regional'),
                mesh_name=d.mesh_name, regional=bool(d.regional),
                dims=list(d.dims), dim_sizes=[int(v) for v in d.dim_sizes],
                units=d.units, history=d.history,
                coords_keys=sorted(d.coords.keys()),
                warnings=[str(w.message) for w in caught])
            store[f'{name}/meta'] = np.array(json.dumps(meta))
            index.append(name)
            print(name, d.mesh_name, d.regional, len(caught), 'warning(s)')
    # create() and get_lat_lon_descriptor()
    created = []
    for tag, args in (('half_degree', (0.5, 0.5)),
                      ('coarse', (10.0, 5.0)),
                      ('regional_box', (1.0, 1.0, 0.0, 30.0, -10.0, 10.0))):
        d = one.get_lat_lon_descriptor(*args)
        store[f'create/{tag}/args'] = np.asarray(args, dtype=float)
        for k in ('lat', 'lon', 'lat_corner', 'lon_corner'):
            store[f'create/{tag}/{k}'] = np.asarray(getattr(d, k))
        store[f'create/{tag}/meta'] = np.array(json.dumps(dict(
            mesh_name=d.mesh_name, regional=bool(d.regional),
            dims=list(d.dims), dim_sizes=[int(v) for v in d.dim_sizes],
            units=d.units)))
        created.append(tag)
        print('create', tag, d.mesh_name, d.regional, d.dim_sizes)
    store['index'] = np.array(json.dumps(dict(read=index, create=created)))
    np.savez_compressed(OUT, **store)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
