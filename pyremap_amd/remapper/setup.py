"""
Materialise the descriptors of a Remapper and validate its configuration:
the behaviour of the reference's ``_setup_remapper``
(``pyremap/remapper/setup.py:5-63``) and of the descriptor factory it calls
(``pyremap/remapper/descriptor.py:21-199``) for the grid kinds that can be
described without pyproj.
"""

from pyremap_amd.descriptor import (
    LatLon2DGridDescriptor,
    LatLonGridDescriptor,
    MpasCellMeshDescriptor,
    MpasEdgeMeshDescriptor,
    MpasVertexMeshDescriptor,
    PointCollectionDescriptor,
    get_lat_lon_descriptor,
)

_TOOL_PREFIX = {'esmf': 'esmf', 'moab': 'mbtr', 'analytic': 'analytic'}
_METHOD_SUFFIX = {'conserve': 'aave', 'bilinear': 'bilin',
                  'neareststod': 'neareststod'}


def _setup_remapper(remapper):
    """Set up the descriptors and check the remapper."""
    descriptors = {}
    for side in ('src', 'dst'):
        descriptor = getattr(remapper, f'{side}_descriptor')
        if descriptor is None:
            info = getattr(remapper, f'{side}_grid_info')
            if 'type' not in info:
                raise ValueError(
                    f'None of the "{side}_from_*()" methods were called')
            descriptor = _get_descriptor(info)
            descriptor.format = remapper.format
        descriptors[side] = descriptor
    src_descriptor, dst_descriptor = descriptors['src'], descriptors['dst']

    map_tool, method = remapper.map_tool, remapper.method
    if remapper.map_filename is None:
        # default name, setup.py:29-42 (a KeyError for unknown tool/method,
        # as in the reference)
        suffix = f'{_TOOL_PREFIX[map_tool]}{_METHOD_SUFFIX[method]}'
        remapper.map_filename = (
            f'map_{src_descriptor.mesh_name}_to_{dst_descriptor.mesh_name}'
            f'_{suffix}.nc')
    if map_tool not in ('moab', 'esmf', 'analytic'):
        raise ValueError(
            f'Unexpected map_tool {map_tool}. Valid '
            f'values are "esmf" or "moab".')
    if isinstance(dst_descriptor, PointCollectionDescriptor) and \
            method not in ('bilinear', 'neareststod'):
        raise ValueError(
            f'method {method} not supported for destination '
            f'grid of type PointCollectionDescriptor.')
    if map_tool == 'moab' and method == 'neareststod':
        raise ValueError('method neareststod not supported by mbtempest.')

    remapper.src_descriptor = src_descriptor
    remapper.dst_descriptor = dst_descriptor


def _get_descriptor(info):
    """Grid-info dict -> descriptor (``remapper/descriptor.py:21-42``)."""
    grid_type = info['type']
    if grid_type == 'mpas':
        return _mpas_descriptor(info)
    if grid_type == 'lon-lat':
        return _lon_lat_descriptor(info)
    if grid_type == 'points':
        return _points_descriptor(info)
    if grid_type == 'proj':
        return _proj_descriptor(info)
    raise ValueError(f'Unexpected grid type {grid_type}')


def _mpas_descriptor(info):
    kinds = {'cell': MpasCellMeshDescriptor, 'edge': MpasEdgeMeshDescriptor,
             'vertex': MpasVertexMeshDescriptor}
    mesh_type = info['mpas_mesh_type']
    if mesh_type not in kinds:
        raise ValueError(f'Unexpected MPAS mesh type {mesh_type}')
    return kinds[mesh_type](info['filename'], mesh_name=info['name'])


def _lon_lat_descriptor(info):
    if 'dlat' in info and 'dlon' in info:
        lon_min = info['lon_min']
        descriptor = get_lat_lon_descriptor(
            dlon=info['dlon'], dlat=info['dlat'], lon_min=lon_min,
            lon_max=lon_min + 360.0)
    else:
        from pyremap_amd.io.netcdf import open_dataset
        ds = open_dataset(info['filename'])
        lon, lat = info['lon'], info['lat']
        ndims = (len(ds[lon].dims), len(ds[lat].dims))
        if ndims not in ((1, 1), (2, 2)):
            raise ValueError(
                f'longitude and latitude coordinates {lon} and {lat} have '
                f'unexpected sizes {ndims[0]} and {ndims[1]}.')
        cls = LatLonGridDescriptor if ndims == (1, 1) else \
            LatLon2DGridDescriptor
        descriptor = cls.read(ds=ds, lon_var_name=lon, lat_var_name=lat,
                              regional=info.get('regional'))
    if 'name' in info:
        descriptor.mesh_name = info['name']
    return descriptor


def _proj_descriptor(info):
    """``remapper/descriptor.py:136-166``: the PROJ string comes from the
    call or from a global attribute of the grid file."""
    from pyremap_amd.descriptor import ProjectionGridDescriptor
    from pyremap_amd.descriptor.projection import projection_from_string
    from pyremap_amd.io.netcdf import open_dataset
    if 'proj_attr' in info:
        proj_str = open_dataset(info['filename']).attrs[info['proj_attr']]
    else:
        proj_str = info['proj_str']
    return ProjectionGridDescriptor.read(
        projection_from_string(proj_str), info['filename'],
        mesh_name=info['name'], x_var_name=info['x'], y_var_name=info['y'])


def _points_descriptor(info):
    from pyremap_amd.io.netcdf import open_dataset
    ds = open_dataset(info['filename'])
    lat_var = ds[info['lat']]
    units = lat_var.attrs.get('units', 'degrees')
    if isinstance(units, bytes):
        units = units.decode()
    return PointCollectionDescriptor(
        lat_var.values, ds[info['lon']].values, info['name'],
        units='radians' if 'rad' in str(units) else 'degrees',
        out_dimension=lat_var.dims[0])
