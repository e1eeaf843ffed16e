"""
pyremap_amd -- MI355X-native weight application behind the pyremap API.

Public names follow ``pyremap/__init__.py:1-31`` of the reference: the
``Remapper`` and the descriptor classes, plus the lightweight
``Dataset`` / ``DataArray`` containers used when xarray is not installed.
"""
from pyremap_amd.descriptor import (  # noqa: F401
    LatLon2DGridDescriptor,
    LatLonGridDescriptor,
    MeshDescriptor,
    MpasCellMeshDescriptor,
    MpasEdgeMeshDescriptor,
    MpasMeshDescriptor,
    MpasVertexMeshDescriptor,
    PointCollectionDescriptor,
    ProjectionGridDescriptor,
    get_lat_lon_descriptor,
)
from pyremap_amd.polar import (  # noqa: F401
    get_polar_descriptor,
    get_polar_descriptor_from_file,
)
from pyremap_amd.remapper import Remapper  # noqa: F401
from pyremap_amd.xr_lite import DataArray, Dataset  # noqa: F401

__version__ = '0.1.0'
