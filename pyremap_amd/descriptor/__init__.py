"""
Mesh / grid descriptors: the attributes the weight-application path reads
(``dims, dim_sizes, coords, mesh_name``; reference
``pyremap/descriptor/mesh_descriptor.py:62-66``) with constructors that do not
need xarray, pyproj or the SCRIP writer.  Writing SCRIP files feeds
ESMF/MOAB weight generation and is out of scope.
"""
from pyremap_amd.descriptor.descriptors import (  # noqa: F401
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
from pyremap_amd.descriptor.projection import (  # noqa: F401
    PolarStereographic,
)
