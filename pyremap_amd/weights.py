"""
Mapping weights without ESMF / MOAB: closed forms
for logically rectangular grids, and ESMF's ``bilinear`` on the dual mesh
when an MPAS mesh is the source.

The reference generates weights only by shelling out to
``ESMF_RegridWeightGen`` or ``mbtempest`` (``pyremap/remapper/build_map.py``),
neither of which exists here.  For grids whose cells are products of two 1-D
axes the common methods have closed forms:

* ``conserve``  -- first-order conservative: S[i, j] = area(dst i n src j) /
  area(dst i).  For lat-lon boxes on the sphere the area factorises into
  (overlap in longitude) x (overlap in sin latitude), for x-y boxes of one
  projection into (overlap in x) x (overlap in y).  ``frac_b`` = covered
  fraction of the destination cell (ESMF's ``destarea`` normalisation: weights
  of a fully covered cell sum to 1).
* ``bilinear``  -- ESMF's construction (:func:`bilinear_3d`): every
  destination point is located in the quad of four neighbouring source cell
  CENTRES, the corners joined by straight lines in 3-D, and takes the patch's
  bilinear weights; a global lat-lon source closes in longitude and is capped
  at either pole by a node that stands for the mean of the adjacent row.
  Reproduces the outputs the reference's tests store (ESMF weights) to the
  rounding of those files.
* ``neareststod`` -- nearest source centre per axis.

* ``bilinear`` FROM an MPAS mesh (its cells, edges or vertices) -- linear
  interpolation on the triangles of the dual mesh, located and weighted along
  straight lines in 3-D, polygons cut into triangles by ESMF's ear-clipping
  rule (:func:`clip_ears`): reproduces the outputs the reference's tests store
  for ``test_mpas_{cell,edge,vertex}_to_latlon`` and
  ``test_mpas_cell_to_stereographic`` to rounding, the unmapped cells
  included.

The result is a :class:`pyremap_amd.io.mapfile.MappingFile` with exactly the
schema ESMF writes (1-based ``row``/``col``, Fortran-ordered grid dims), so it
goes through the same ``_load_mapping`` as any other mapping file.
"""
import numpy as np

from pyremap_amd.descriptor import (
    LatLonGridDescriptor,
    MpasMeshDescriptor,
    PointCollectionDescriptor,
    ProjectionGridDescriptor,
)
from pyremap_amd.io.mapfile import MappingFile

METHODS = ('conserve', 'bilinear', 'neareststod')


# ---------------------------------------------------------------------------
# one axis at a time: sparse (dst index, src index, weight) triplets
# ---------------------------------------------------------------------------

def _ascending(edges):
    """(edges in ascending order, permutation of the cells)"""
    edges = np.asarray(edges, dtype=np.float64)
    n = len(edges) - 1
    if edges[-1] >= edges[0]:
        return edges, np.arange(n)
    return edges[::-1].copy(), np.arange(n)[::-1].copy()


def overlap_1d(src_edges, dst_edges, period=None):
    """
    Lengths of the pairwise overlaps of two sets of consecutive intervals.
    Returns (j, i, length): destination cell, source cell, overlap > 0.
    With ``period`` the source intervals repeat every ``period`` (longitude).
    """
    se, sperm = _ascending(src_edges)
    de, dperm = _ascending(dst_edges)
    ns = len(se) - 1
    if period is not None:
        # enough copies of the source axis to cover the destination extent
        lo = int(np.floor((de[0] - se[-1]) / period))
        hi = int(np.ceil((de[-1] - se[0]) / period))
        shifts = np.arange(lo, hi + 1) * period
    else:
        shifts = np.zeros(1)
    out_j, out_i, out_len = [], [], []
    for shift in shifts:
        lo_e = se[:-1] + shift
        hi_e = se[1:] + shift
        # destination cells a source interval can touch
        first = np.searchsorted(de, lo_e, side='right') - 1
        last = np.searchsorted(de, hi_e, side='left') - 1
        first = np.clip(first, 0, len(de) - 2)
        last = np.clip(last, -1, len(de) - 2)
        count = np.maximum(last - first + 1, 0)
        i = np.repeat(np.arange(ns), count)
        start = np.repeat(first, count)
        offs = np.arange(count.sum()) - np.repeat(
            np.cumsum(count) - count, count)
        j = start + offs
        length = np.minimum(hi_e[i], de[j + 1]) - np.maximum(lo_e[i], de[j])
        keep = length > 0.0
        out_j.append(dperm[j[keep]])
        out_i.append(sperm[i[keep]])
        out_len.append(length[keep])
    j = np.concatenate(out_j)
    i = np.concatenate(out_i)
    length = np.concatenate(out_len)
    order = np.lexsort((i, j))
    return j[order], i[order], length[order]


def nearest_1d(src_centres, dst_points, period=None):
    sc = np.asarray(src_centres, dtype=np.float64)
    dp = np.asarray(dst_points, dtype=np.float64)
    diff = dp[:, None] - sc[None, :]
    if period is not None:
        diff = np.mod(diff + 0.5 * period, period) - 0.5 * period
    i = np.abs(diff).argmin(axis=1)
    j = np.arange(len(dp))
    return j, i, np.ones(len(dp))


def _tensor(ay, ax, ny_src, nx_src, ny_dst, nx_dst):
    """Kronecker product of the per-axis triplets -> 2-D (row, col, S)."""
    jy, iy, wy = ay
    jx, ix, wx = ax
    row = (jy[:, None] * nx_dst + jx[None, :]).reshape(-1)
    col = (iy[:, None] * nx_src + ix[None, :]).reshape(-1)
    S = (wy[:, None] * wx[None, :]).reshape(-1)
    order = np.lexsort((col, row))
    return row[order], col[order], S[order]


# ---------------------------------------------------------------------------
# descriptors -> axes
# ---------------------------------------------------------------------------

def _axes(descriptor):
    """(y centres, x centres, y edges, x edges, periodic period or None,
    'sphere' | 'plane') of a rectangular descriptor."""
    if isinstance(descriptor, LatLonGridDescriptor):
        scale = 1.0 if 'rad' in descriptor.units else np.pi / 180.0
        lat = np.asarray(descriptor.lat) * scale
        lon = np.asarray(descriptor.lon) * scale
        lat_e = np.clip(np.asarray(descriptor.lat_corner) * scale,
                        -0.5 * np.pi, 0.5 * np.pi)
        lon_e = np.asarray(descriptor.lon_corner) * scale
        period = None if descriptor.regional else 2.0 * np.pi
        return lat, lon, lat_e, lon_e, period, 'sphere'
    if isinstance(descriptor, ProjectionGridDescriptor):
        return (np.asarray(descriptor.y), np.asarray(descriptor.x),
                np.asarray(descriptor.y_corner),
                np.asarray(descriptor.x_corner), None, 'plane')
    raise TypeError(
        f'analytic weights need a LatLonGridDescriptor or a '
        f'ProjectionGridDescriptor, not {type(descriptor).__name__}')


def _points(descriptor):
    """(lat, lon) in radians of a point-like destination, or None."""
    if isinstance(descriptor, PointCollectionDescriptor):
        scale = 1.0 if 'rad' in descriptor.units else np.pi / 180.0
        return (np.asarray(descriptor.lat, dtype=np.float64) * scale,
                np.asarray(descriptor.lon, dtype=np.float64) * scale)
    if isinstance(descriptor, MpasMeshDescriptor) and descriptor.coords:
        c = descriptor.coords
        return (np.asarray(c[descriptor._lat_coord]['data'], np.float64),
                np.asarray(c[descriptor._lon_coord]['data'], np.float64))
    return None


def _forward(projection, lon_deg, lat_deg):
    """(x, y) of points given in degrees, through this package's projection
    class or a ``pyproj.Proj`` (which is callable that way)."""
    if hasattr(projection, 'forward'):
        return projection.forward(lon_deg, lat_deg)
    return projection(lon_deg, lat_deg)


def _to_points(src_descriptor, plat, plon, dst_dims, method):
    """
    Rectangular grid -> points given by latitude / longitude in radians (MPAS
    cell centres, point collections, or the cell centres of a grid of another
    kind).  ``bilinear``: :func:`bilinear_3d` (ESMF's way); points no quad of
    source centres -- or pole cap of a global source -- holds are not mapped.
    ``neareststod``: the nearest centre per axis; points outside the source
    cells are not mapped.
    """
    if method == 'conserve':
        raise ValueError(
            'method conserve needs cells of the same kind of grid on both '
            'sides; towards points or across grid kinds only bilinear and '
            'neareststod have a closed form')
    sy, sx, sye, sxe, period, kind = _axes(src_descriptor)
    n = len(plat)
    if method == 'bilinear':
        row, col, S, mapped = bilinear_3d(src_descriptor, plat, plon)
        row, col, S = _merged(row, col, S)
        return MappingFile(
            len(sy) * len(sx), n,
            np.array([len(sx), len(sy)], dtype=np.int32),
            np.asarray(dst_dims, dtype=np.int32),
            (row + 1).astype(np.int32), (col + 1).astype(np.int32), S,
            mapped.astype(np.float64))
    if kind == 'sphere':
        py, px = plat, plon
    else:
        if src_descriptor.projection is None:
            raise ValueError('the source grid has no projection to locate '
                             'the destination points with')
        px, py = _forward(src_descriptor.projection, np.degrees(plon),
                          np.degrees(plat))
    axis = nearest_1d
    jy, iy, wy = axis(sy, py)
    jx, ix, wx = axis(sx, px, period)
    # pair every y entry of a point with every x entry of the same point
    oy = np.argsort(jy, kind='stable')
    ox = np.argsort(jx, kind='stable')
    jy, iy, wy = jy[oy], iy[oy], wy[oy]
    jx, ix, wx = jx[ox], ix[ox], wx[ox]
    cy = np.bincount(jy, minlength=n)
    cx = np.bincount(jx, minlength=n)
    sy0 = np.cumsum(cy) - cy
    sx0 = np.cumsum(cx) - cx
    per = cy * cx
    point = np.repeat(np.arange(n), per)
    k = np.arange(per.sum()) - np.repeat(np.cumsum(per) - per, per)
    ky = sy0[point] + k // np.maximum(cx[point], 1)
    kx = sx0[point] + k % np.maximum(cx[point], 1)
    row = point
    col = iy[ky] * len(sx) + ix[kx]
    S = wy[ky] * wx[kx]
    # what counts as "inside" for nearest: inside the outermost cells; a
    # global lat-lon source has no longitude limits and its latitude rows
    # reach the poles
    ylim, xlim = (sye[0], sye[-1]), (sxe[0], sxe[-1])
    inside = np.ones(n, dtype=bool)
    if period is None:
        inside &= (px >= min(xlim)) & (px <= max(xlim))
    if period is None or kind == 'plane':
        inside &= (py >= min(ylim)) & (py <= max(ylim))
    keep = inside[row]
    row, col, S = row[keep], col[keep], S[keep]
    frac_b = inside.astype(np.float64)
    order = np.lexsort((col, row))
    return MappingFile(
        len(sy) * len(sx), n, np.array([len(sx), len(sy)], dtype=np.int32),
        np.asarray(dst_dims, dtype=np.int32),
        (row[order] + 1).astype(np.int32), (col[order] + 1).astype(np.int32),
        S[order], frac_b)


# ---------------------------------------------------------------------------
# an MPAS mesh as the SOURCE: linear interpolation on the dual mesh
# ---------------------------------------------------------------------------

def _unit(lat, lon):
    lat, lon = np.broadcast_arrays(lat, lon)
    return np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon),
                     np.sin(lat)], axis=-1)


def clip_ears(xyz, poly, count):
    """
    Triangulate convex polygons the way ESMF does before it interpolates on
    elements with more than four corners: repeatedly cut off the corner whose
    two edge vectors have the LARGEST dot product (3-D Cartesian, not
    normalised) until a triangle is left.  (Inferred from the outputs the
    reference's tests store for `test_mpas_vertex_to_latlon` and
    `test_mpas_edge_to_latlon`: of the 14 triangulations of each of the 7 088
    hexagons that the stored values pin down, this rule picks the one ESMF
    used, every time; fans and the other greedy measures -- angle, area,
    diagonal -- do not.)

    ``poly``: (n_poly, max_corners) node ids, the first ``count[i]`` of row i
    valid, in order around the polygon (either orientation).  Returns the
    triangles, (nt, 3) node ids.
    """
    poly = np.array(poly, dtype=np.int64)
    count = np.array(count, dtype=np.int64)
    keep = count >= 3
    poly, count = poly[keep], count[keep]
    out = []
    width = poly.shape[1] if len(poly) else 0
    slots = np.arange(width)
    while len(poly):
        done = count == 3
        if done.any():
            out.append(poly[done][:, :3])
            poly, count = poly[~done], count[~done]
            if not len(poly):
                break
        valid = slots[None, :] < count[:, None]
        rows = np.arange(len(poly))[:, None]
        prev = poly[rows, (slots[None, :] - 1) % count[:, None]]
        nxt = poly[rows, (slots[None, :] + 1) % count[:, None]]
        here = xyz[np.where(valid, poly, 0)]
        dot = ((xyz[np.where(valid, prev, 0)] - here) *
               (xyz[np.where(valid, nxt, 0)] - here)).sum(axis=-1)
        dot[~valid] = -np.inf
        ear = dot.argmax(axis=1)
        r = np.arange(len(poly))
        out.append(np.stack([prev[r, ear], poly[r, ear], nxt[r, ear]],
                            axis=1))
        # delete the ear: shift the corners behind it one slot left
        shift = slots[None, :] >= ear[:, None]
        poly = np.where(shift, poly[rows, np.minimum(slots + 1,
                                                     width - 1)[None, :]],
                        poly)
        count = count - 1
    return np.concatenate(out) if out else np.zeros((0, 3), dtype=np.int64)


def _dual_triangles(descriptor):
    """
    The mesh ESMF interpolates on when an MPAS mesh is the source of a
    ``bilinear`` map: the DUAL of the SCRIP cells the reference writes for it
    -- nodes at the cells' centres, one element around every corner three or
    more SCRIP cells share -- with elements of more than three corners cut
    into triangles (:func:`clip_ears`):

    * cells (``mpas_cell_mesh_descriptor.py:84-167``: corners = vertices):
      the triangle of the three cell centres around every vertex;
    * vertices (``mpas_vertex_mesh_descriptor.py:107-180``: corners = cell
      centres and edge midpoints): the polygon of the vertices around every
      cell;
    * edges (``mpas_edge_mesh_descriptor.py:106-190``: corners = the two
      vertices and the two cell centres): the triangle of the three edge
      midpoints around every vertex and the polygon of the edge midpoints
      around every cell.

    Elements at the boundary of the mesh that lack a member (a land cell)
    do not exist: destination points there stay unmapped.  Returns ``(xyz of
    the nodes (n, 3), triangles (nt, 3) of 0-based node ids)``.
    """
    if getattr(descriptor, 'filename', None) is None:
        raise ValueError(
            'weights FROM an MPAS mesh need its mesh file (connectivity): '
            'construct the descriptor with filename=')
    from pyremap_amd.io.netcdf import open_dataset
    kind = descriptor._dim
    wanted = {'nCells': ['latCell', 'lonCell', 'cellsOnVertex'],
              'nVertices': ['latVertex', 'lonVertex', 'verticesOnCell',
                            'nEdgesOnCell'],
              'nEdges': ['latEdge', 'lonEdge', 'edgesOnVertex',
                         'cellsOnEdge', 'latCell', 'lonCell']}[kind]
    ds = open_dataset(descriptor.filename, variables=wanted)
    xyz = _unit(np.asarray(ds[wanted[0]].values, dtype=np.float64),
                np.asarray(ds[wanted[1]].values, dtype=np.float64))
    n = len(xyz)

    def triples(name):
        t = np.asarray(ds[name].values, dtype=np.int64)
        if t.ndim != 2 or t.shape[1] != 3:
            raise ValueError(f'{name}: a vertexDegree of 3 is needed')
        return t[((t > 0) & (t <= n)).all(axis=1)] - 1

    def polygons(name):
        poly = np.asarray(ds[name].values, dtype=np.int64) - 1
        count = np.asarray(ds['nEdgesOnCell'].values, dtype=np.int64)
        whole = ((poly >= 0) & (poly < n)) | \
            (np.arange(poly.shape[1])[None, :] >= count[:, None])
        ok = whole.all(axis=1)
        return clip_ears(xyz, poly[ok], count[ok])

    def edges_around_cells():
        # (the reference's edge descriptor needs cellsOnEdge only, and mesh
        # files cut down to what it reads carry no edgesOnCell: the edges of
        # a cell in order of their bearing from the cell centre)
        coe = np.asarray(ds['cellsOnEdge'].values, dtype=np.int64) - 1
        lat = np.asarray(ds['latCell'].values, dtype=np.float64)
        lon = np.asarray(ds['lonCell'].values, dtype=np.float64)
        edge = np.repeat(np.arange(len(coe)), 2)
        cell = coe.reshape(-1)
        keep = (cell >= 0) & (cell < len(lat))
        edge, cell = edge[keep], cell[keep]
        east = np.stack([-np.sin(lon), np.cos(lon), np.zeros_like(lon)], -1)
        north = np.stack([-np.sin(lat) * np.cos(lon),
                          -np.sin(lat) * np.sin(lon), np.cos(lat)], -1)
        off = xyz[edge] - _unit(lat, lon)[cell]
        bearing = np.arctan2((off * north[cell]).sum(-1),
                             (off * east[cell]).sum(-1))
        order = np.lexsort((bearing, cell))
        edge, cell = edge[order], cell[order]
        count = np.bincount(cell, minlength=len(lat))
        start = np.cumsum(count) - count
        poly = np.zeros((len(lat), max(int(count.max()), 3)), dtype=np.int64)
        poly[cell, np.arange(len(cell)) - start[cell]] = edge
        return clip_ears(xyz, poly, count)

    if kind == 'nCells':
        tri = triples('cellsOnVertex')
    elif kind == 'nVertices':
        tri = polygons('verticesOnCell')
    else:
        tri = np.concatenate([triples('edgesOnVertex'),
                              edges_around_cells()])
    return xyz, tri


def locate_in_triangles(xyz, tri, points, tol=1e-12, chunk=1 << 18):
    """
    For every unit vector in ``points`` the spherical triangle (corners
    ``xyz[tri]``) that holds it, with the weights of its corners: the
    barycentric coordinates of the point's central projection onto the
    triangle's plane (straight lines in 3-D -- ESMF's default ``cartesian``
    line type for bilinear).  Returns ``(triangle index or -1, weights (n,
    3))``.  A uniform hash grid over the triangle centroids, one cell as wide
    as the longest triangle edge, bounds the candidates to the 27 cells
    around the point.
    """
    corners = xyz[tri]                                    # (nt, 3, 3)
    inv = np.linalg.inv(np.transpose(corners, (0, 2, 1)))  # columns a, b, c
    cent = corners.sum(axis=1)
    cent /= np.linalg.norm(cent, axis=1)[:, None]
    edge = max(np.linalg.norm(corners[:, i] - corners[:, (i + 1) % 3],
                              axis=1).max() for i in range(3))
    h = float(min(max(edge, 1e-6), 2.0))
    nb = int(np.ceil(2.0 / h)) + 2

    def cell_of(p):
        return np.floor((p + 1.0) / h).astype(np.int64) + 1

    ck = cell_of(cent)
    key = (ck[:, 0] * nb + ck[:, 1]) * nb + ck[:, 2]
    order = np.argsort(key, kind='stable')
    skey = key[order]
    n = len(points)
    found = np.full(n, -1, dtype=np.int64)
    weights = np.zeros((n, 3))
    offsets = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1)
               for c in (-1, 0, 1)]
    for c0 in range(0, n, chunk):
        q = points[c0:c0 + chunk]
        qc = cell_of(q)
        best = np.full(len(q), len(tri), dtype=np.int64)
        for off in offsets:
            k = ((qc[:, 0] + off[0]) * nb + qc[:, 1] + off[1]) * nb + \
                qc[:, 2] + off[2]
            lo = np.searchsorted(skey, k, side='left')
            cnt = np.searchsorted(skey, k, side='right') - lo
            total = int(cnt.sum())
            if total == 0:
                continue
            qi = np.repeat(np.arange(len(q)), cnt)
            ti = order[np.repeat(lo, cnt) + np.arange(total) -
                       np.repeat(np.cumsum(cnt) - cnt, cnt)]
            w = np.einsum('nij,nj->ni', inv[ti], q[qi])
            tot = w.sum(axis=1)
            with np.errstate(divide='ignore', invalid='ignore'):
                w = w / tot[:, None]
            inside = (tot > 0.0) & (w >= -tol).all(axis=1)
            # one triangle per point, the lowest index (a point on a shared
            # edge gets the same value from either side)
            np.minimum.at(best, qi[inside], ti[inside])
        hit = best < len(tri)
        t = best[hit]
        w = np.einsum('nij,nj->ni', inv[t], q[hit])
        w = np.clip(w / w.sum(axis=1)[:, None], 0.0, None)
        weights[c0:c0 + chunk][hit] = w / w.sum(axis=1)[:, None]
        found[c0:c0 + chunk][hit] = t
    return found, weights


def _from_cell_mesh(src_descriptor, plat, plon, dst_dims, method):
    """MPAS cells / edges / vertices -> points (radians): ``bilinear`` = linear
    on the triangles of the dual mesh; destination points no triangle holds (land, the gaps at the
    mesh boundary) stay unmapped, ``frac_b`` = 0, as ESMF leaves them."""
    if method != 'bilinear':
        raise ValueError(
            f'from an MPAS mesh only bilinear has a closed form here, not '
            f'{method!r} (conservative weights need polygon clipping: ESMF / '
            f'MOAB)')
    xyz, tri = _dual_triangles(src_descriptor)
    found, w = locate_in_triangles(xyz, tri, _unit(plat, plon))
    hit = np.nonzero(found >= 0)[0]
    row = np.repeat(hit, 3)
    col = tri[found[hit]].reshape(-1)
    S = w[hit].reshape(-1)
    order = np.lexsort((col, row))
    frac_b = (found >= 0).astype(np.float64)
    return MappingFile(
        len(xyz), len(plat), np.array([len(xyz)], dtype=np.int32),
        np.asarray(dst_dims, dtype=np.int32),
        (row[order] + 1).astype(np.int32), (col[order] + 1).astype(np.int32),
        S[order], frac_b)



# ---------------------------------------------------------------------------
# bilinear from a logically rectangular grid, as ESMF does it: on the quads
# between four neighbouring cell centres, along straight lines in 3-D
# ---------------------------------------------------------------------------

def _solve_quads(P, q, iters=12):
    """
    Where the ray from the sphere's centre through each unit vector ``q[n]``
    meets the bilinear patch through the four corners ``P[n]`` (in the order
    (-1, -1), (1, -1), (1, 1), (-1, 1) of the patch coordinates): Newton on
    ``sum_i N_i(s, t) P_i - r q = 0``.  Returns ``(s, t)``; NaN where the
    iteration left the neighbourhood of the patch.
    """
    n = len(q)
    s = np.zeros(n)
    t = np.zeros(n)
    r = np.ones(n)
    p0, p1, p2, p3 = P[:, 0], P[:, 1], P[:, 2], P[:, 3]
    # X(s, t) = c0 + s c1 + t c2 + s t c3
    c0 = 0.25 * (p0 + p1 + p2 + p3)
    c1 = 0.25 * (-p0 + p1 + p2 - p3)
    c2 = 0.25 * (-p0 - p1 + p2 + p3)
    c3 = 0.25 * (p0 - p1 + p2 - p3)
    last = False
    with np.errstate(all='ignore'):
        for _ in range(iters):
            F = c0 + s[:, None] * c1 + t[:, None] * c2 + \
                (s * t)[:, None] * c3 - r[:, None] * q
            a = c1 + t[:, None] * c3           # dX/ds
            b = c2 + s[:, None] * c3           # dX/dt
            # J = [a, b, -q]; Cramer's rule on J d = -F
            bq = np.cross(b, q)
            det = -(a * bq).sum(axis=1)
            d0 = (F * bq).sum(axis=1) / det
            d1 = (a * np.cross(F, q)).sum(axis=1) / det
            d2 = -(a * np.cross(b, F)).sum(axis=1) / det
            s, t, r = s + d0, t + d1, r + d2
            far = ~(np.abs(s) <= 50.0) | ~(np.abs(t) <= 50.0)
            s[far] = t[far] = np.nan
            r[far] = 1.0
            if last:
                break
            # quadratic convergence: one more step after 1e-8 is rounding
            step = np.maximum(np.abs(d0), np.abs(d1))
            last = not (step[~far] > 1e-8).any()
    return s, t


def _merged(row, col, S):
    """Triplets sorted by (row, col), duplicates summed."""
    order = np.lexsort((col, row))
    row, col, S = row[order], col[order], S[order]
    if len(row) == 0:
        return row, col, S
    head = np.ones(len(row), dtype=bool)
    head[1:] = (row[1:] != row[:-1]) | (col[1:] != col[:-1])
    start = np.nonzero(head)[0]
    return row[start], col[start], np.add.reduceat(S, start)


def _grid_nodes(descriptor):
    """
    The nodes ESMF's bilinear works on for a rectangular source grid -- its
    cell CENTRES, ``(ny, nx)`` -- as unit vectors, with: whether the columns
    close around the globe, whether the first / last row is capped by a pole
    (ESMF's default for a global source: an artificial node at the pole whose
    value is the mean of the row next to it), and a function that guesses the
    quad ``(j, i)`` holding points given by latitude / longitude in radians.
    """
    if isinstance(descriptor, LatLonGridDescriptor):
        scale = 1.0 if 'rad' in descriptor.units else np.pi / 180.0
        lat = np.asarray(descriptor.lat, dtype=np.float64) * scale
        lon = np.asarray(descriptor.lon, dtype=np.float64) * scale
        glob = not descriptor.regional
        nodes = _unit(lat[:, None], lon[None, :])

        def guess(plat, plon):
            return _bracket(lat, plat, None), \
                _bracket(lon, plon, 2.0 * np.pi if glob else None)
        return nodes, glob, glob, guess
    if isinstance(descriptor, ProjectionGridDescriptor):
        x = np.asarray(descriptor.x, dtype=np.float64)
        y = np.asarray(descriptor.y, dtype=np.float64)
        xx, yy = np.meshgrid(x, y)
        lat, lon = descriptor.project_to_lat_lon(xx, yy)
        if lat is None:
            raise ValueError('the source grid has no usable projection')
        nodes = _unit(np.radians(lat), np.radians(lon))

        def guess(plat, plon):
            px, py = _forward(descriptor.projection, np.degrees(plon),
                              np.degrees(plat))
            return _bracket(y, py, None), _bracket(x, px, None)
        return nodes, False, False, guess
    raise TypeError(
        f'analytic weights need a LatLonGridDescriptor or a '
        f'ProjectionGridDescriptor, not {type(descriptor).__name__}')


def _bracket(axis, p, period):
    """Index k of the interval [axis[k], axis[k + 1]] that holds p (axis
    ascending or descending; with ``period`` the last interval closes the
    circle); clipped into range."""
    n = len(axis)
    if n < 2:
        return np.zeros(len(p), dtype=np.int64)
    flip = axis[-1] < axis[0]
    a = axis[::-1] if flip else axis
    if period is not None:
        t = a[0] + np.mod(p - a[0], period)
        k = np.clip(np.searchsorted(a, t, side='right') - 1, 0, n - 1)
        return (n - 2 - k) % n if flip else k
    k = np.clip(np.searchsorted(a, p, side='right') - 1, 0, n - 2)
    return n - 2 - k if flip else k


def bilinear_3d(src_descriptor, plat, plon, tol=1e-10, chunk=1 << 20):
    """
    ESMF's ``bilinear`` from a rectangular grid to points (radians): each
    point is located in a quad of four neighbouring source centres -- corners
    joined by straight lines in 3-D, the point carried onto the patch along
    the ray from the sphere's centre -- and takes the patch's bilinear
    weights; a global lat-lon source closes around the globe and is capped
    at either pole by triangles to a pole node that stands for the mean of
    the adjacent row.  (Agrees with the outputs the reference's tests store,
    made with ESMF weights, to the float32 rounding of those files; bilinear
    interpolation in latitude / longitude, which this replaced, differs from
    them by up to 5e-3 K on the 1-degree SST file.)

    Returns ``(row, col, S, mapped)``: 0-based triplets and the mask of the
    points some quad or cap holds.
    """
    plat = np.asarray(plat, dtype=np.float64)
    plon = np.asarray(plon, dtype=np.float64)
    if len(plat) > chunk:
        # bounded memory whatever the destination grid (30 M points for a
        # 1 km Antarctic grid): the points in pieces
        parts = [bilinear_3d(src_descriptor, plat[c:c + chunk],
                             plon[c:c + chunk], tol, chunk)
                 for c in range(0, len(plat), chunk)]
        return (np.concatenate([p[0] + k * chunk
                                for k, p in enumerate(parts)]),
                np.concatenate([p[1] for p in parts]),
                np.concatenate([p[2] for p in parts]),
                np.concatenate([p[3] for p in parts]))
    nodes, periodic, capped, guess = _grid_nodes(src_descriptor)
    ny, nx = nodes.shape[:2]
    q = _unit(plat, plon)
    n = len(q)
    j0, i0 = guess(plat, plon)
    mapped = np.zeros(n, dtype=bool)
    idx = np.zeros((n, 4), dtype=np.int64)
    wgt = np.zeros((n, 4))
    if ny >= 2 and nx >= 2:
        todo = np.arange(n)
        for dj, di in ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1),
                       (-1, 1), (1, -1), (1, 1)):
            if not len(todo):
                break
            j = j0[todo] + dj
            i = i0[todo] + di
            if periodic:
                i = i % nx
                i1 = (i + 1) % nx
                ok = (j >= 0) & (j <= ny - 2)
            else:
                i1 = i + 1
                ok = (j >= 0) & (j <= ny - 2) & (i >= 0) & (i <= nx - 2)
            j, i, i1, pts = j[ok], i[ok], i1[ok], todo[ok]
            P = np.stack([nodes[j, i], nodes[j, i1], nodes[j + 1, i1],
                          nodes[j + 1, i]], axis=1)
            s, t = _solve_quads(P, q[pts])
            with np.errstate(invalid='ignore'):
                inside = (np.abs(s) <= 1.0 + tol) & (np.abs(t) <= 1.0 + tol)
            s = np.clip(s[inside], -1.0, 1.0)
            t = np.clip(t[inside], -1.0, 1.0)
            hit = pts[inside]
            wgt[hit] = np.stack([(1 - s) * (1 - t), (1 + s) * (1 - t),
                                 (1 + s) * (1 + t), (1 - s) * (1 + t)],
                                -1) * 0.25
            idx[hit] = np.stack([j * nx + i, j * nx + i1, (j + 1) * nx + i1,
                                 (j + 1) * nx + i], -1)[inside]
            mapped[hit] = True
            todo = todo[~mapped[todo]]
    row = np.repeat(np.nonzero(mapped)[0], 4)
    col = idx[mapped].reshape(-1)
    S = wgt[mapped].reshape(-1)
    if capped and (~mapped).any():
        # the pole caps: triangles (pole, node i, node i + 1) of the first /
        # last row; the pole's share goes to the whole row in equal parts
        rest = np.nonzero(~mapped)[0]
        for jrow in (0, ny - 1):
            if not len(rest):
                break
            zs = nodes[jrow, :, 2].mean()
            pole = np.array([0.0, 0.0, 1.0 if zs > 0 else -1.0])
            for di in (0, -1, 1):
                if not len(rest):
                    break
                i = (i0[rest] + di) % nx
                i1 = (i + 1) % nx
                M = np.stack([np.broadcast_to(pole, (len(rest), 3)),
                              nodes[jrow, i], nodes[jrow, i1]], -1)
                w = np.linalg.solve(M, q[rest][:, :, None])[:, :, 0]
                tot = w.sum(axis=1)
                w = w / tot[:, None]
                inside = (tot > 0) & (w >= -tol).all(axis=1)
                hit = rest[inside]
                w = np.clip(w[inside], 0.0, None)
                w = w / w.sum(axis=1)[:, None]
                ring = jrow * nx + np.arange(nx)
                row = np.concatenate([row, np.repeat(hit, nx + 2)])
                col = np.concatenate([col, np.concatenate(
                    [np.broadcast_to(ring, (len(hit), nx)),
                     (jrow * nx + i[inside])[:, None],
                     (jrow * nx + i1[inside])[:, None]], axis=1).reshape(-1)])
                S = np.concatenate([S, np.concatenate(
                    [np.repeat(w[:, :1] / nx, nx, axis=1), w[:, 1:]],
                    axis=1).reshape(-1)])
                mapped[hit] = True
                rest = rest[~mapped[rest]]
    keep = S != 0.0
    return row[keep], col[keep], S[keep], mapped



def _cell_centres(descriptor):
    """(lat, lon) in radians of every cell centre of a rectangular grid, in
    C order, and its Fortran-ordered dims."""
    if isinstance(descriptor, LatLonGridDescriptor):
        scale = 1.0 if 'rad' in descriptor.units else np.pi / 180.0
        lat, lon = np.meshgrid(np.asarray(descriptor.lat) * scale,
                               np.asarray(descriptor.lon) * scale,
                               indexing='ij')
    else:
        xx, yy = np.meshgrid(descriptor.x, descriptor.y)
        lat, lon = descriptor.project_to_lat_lon(xx, yy)
        if lat is None:
            raise ValueError('the destination grid has no usable projection')
        lat, lon = np.radians(lat), np.radians(lon)
    return lat.reshape(-1), lon.reshape(-1), \
        [lat.shape[1], lat.shape[0]]


def build_weights(src_descriptor, dst_descriptor, method='conserve'):
    """
    The mapping between two rectangular grids (lat-lon or on a map
    projection; ``conserve`` only between grids of the same kind), or from
    one to scattered points (an MPAS mesh's cell / edge / vertex positions,
    a point collection), as a :class:`MappingFile`.
    """
    if method not in METHODS:
        raise ValueError(f'method {method!r}: expected one of {METHODS}')
    points = _points(dst_descriptor)
    if isinstance(src_descriptor, MpasMeshDescriptor):
        if points is not None:
            return _from_cell_mesh(src_descriptor, points[0], points[1],
                                   [len(points[0])], method)
        lat, lon, dims = _cell_centres(dst_descriptor)
        return _from_cell_mesh(src_descriptor, lat, lon, dims, method)
    if points is not None:
        return _to_points(src_descriptor, points[0], points[1],
                          [len(points[0])], method)
    if method == 'bilinear':
        # the destination cell centres are points for the source grid
        lat, lon, dims = _cell_centres(dst_descriptor)
        return _to_points(src_descriptor, lat, lon, dims, method)
    sy, sx, sye, sxe, period, kind = _axes(src_descriptor)
    dy, dx, dye, dxe, _, dkind = _axes(dst_descriptor)
    same_kind = kind == dkind
    if same_kind and kind == 'plane':
        ps, pd = src_descriptor.projection, dst_descriptor.projection
        same_kind = ps is pd or \
            getattr(ps, 'srs', ps) == getattr(pd, 'srs', pd)
    if not same_kind:
        # across grid kinds (projection <-> lat-lon, two projections): the
        # destination cell centres are points for the source grid
        lat, lon, dims = _cell_centres(dst_descriptor)
        return _to_points(src_descriptor, lat, lon, dims, method)
    ny_s, nx_s, ny_d, nx_d = len(sy), len(sx), len(dy), len(dx)
    if method == 'conserve':
        if kind == 'sphere':
            sye, dye = np.sin(sye), np.sin(dye)     # area ~ dlon * dsin(lat)
        jy, iy, ly = overlap_1d(sye, dye)
        jx, ix, lx = overlap_1d(sxe, dxe, period)
        wy = ly / np.abs(np.diff(dye))[jy]
        wx = lx / np.abs(np.diff(dxe))[jx]
        row, col, S = _tensor((jy, iy, wy), (jx, ix, wx), ny_s, nx_s, ny_d,
                              nx_d)
        frac_b = np.bincount(row, weights=S, minlength=ny_d * nx_d)
        frac_b = np.minimum(frac_b, 1.0)
    else:
        row, col, S = _tensor(nearest_1d(sy, dy), nearest_1d(sx, dx, period),
                              ny_s, nx_s, ny_d, nx_d)
        frac_b = np.ones(ny_d * nx_d)
        if period is None:
            # destination points outside the source cells are not mapped
            ylim, xlim = (sye[0], sye[-1]), (sxe[0], sxe[-1])
            inside_y = (dy >= min(ylim)) & (dy <= max(ylim))
            if kind == 'sphere':
                # latitude rows reach the poles: nothing is outside
                inside_y = (dy >= min(sye[0], sye[-1])) & \
                    (dy <= max(sye[0], sye[-1]))
            inside_x = (dx >= min(xlim)) & (dx <= max(xlim))
            inside = (inside_y[:, None] & inside_x[None, :]).reshape(-1)
            keep = inside[row]
            row, col, S = row[keep], col[keep], S[keep]
            frac_b = inside.astype(np.float64)
    return MappingFile(
        ny_s * nx_s, ny_d * nx_d,
        np.array([nx_s, ny_s], dtype=np.int32),
        np.array([nx_d, ny_d], dtype=np.int32),
        (row + 1).astype(np.int32), (col + 1).astype(np.int32), S, frac_b)


def write_weights(filename, src_descriptor, dst_descriptor,
                  method='conserve'):
    """Build the weights and write them as a mapping file."""
    from pyremap_amd.io.mapfile import write_mapping
    m = build_weights(src_descriptor, dst_descriptor, method)
    write_mapping(filename, m.n_a, m.n_b, m.src_grid_dims, m.dst_grid_dims,
                  m.row, m.col, m.S, m.frac_b,
                  attrs={'map_method': method,
                         'weight_generator': 'pyremap_amd.weights (analytic)',
                         'normalization': 'destarea',
                         'domain_a': str(src_descriptor.mesh_name),
                         'domain_b': str(dst_descriptor.mesh_name)})
    return m
