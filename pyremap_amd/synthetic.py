"""
Seeded synthetic mapping data with the shapes of BASELINE.json's configs.

pyremap never computes weights itself -- ``build_map()`` shells out to
ESMF / MOAB (``pyremap/remapper/build_map.py:47-91``), neither of which
exists here -- so benchmarks and tests use generated triplets that are
faithful in shape: sizes, entries per row, 1-based unsorted (row, col, S),
``frac_b``, Fortran-ordered grid dims.

Generators run in torch so the large configs (1e8 entries) can be produced
on the GPU; with ``device='cpu'`` they serve the CPU tests.

* :func:`bilinear_map`     exact tensor-product bilinear weights between two
  regular 2-D grids (4 entries per row, rows sum to 1, ``frac_b = 1``):
  configs 1 and 4.
* :func:`conservative_map` overlap-like weights: each destination cell draws
  k ~ U{lo..hi} distinct source cells from the 2-D neighbourhood of its
  position in a virtual raster numbering of the source mesh (so rows that
  are neighbours in either grid direction share source cells, as real
  overlaps do), positive (or signed, for 2nd-order) weights scaled so that a
  row sums to ``frac_b``; a blocky "land" mask leaves rows empty with
  ``frac_b = 0``; every source cell the draws missed is then given to the
  destination cell that contains it, so -- like a real overlap map -- no
  source cell goes unread (``cover``): configs 2, 3, 5 and the north-star
  headline H.  ``locality='mesh'`` keeps that geometry and RENUMBERS the
  source cells the way an MPAS mesh numbers its cells (:func:`mesh_numbering`)
  -- no real mesh is numbered along the destination raster;
  ``locality='scatter'`` renumbers them at random (no id locality at all).
* :func:`knn_map`          overlap-like weights from REAL cell centres (the
  QU240 mesh of the reference's test fixtures) to a regular lat-lon grid.
"""

CONFIGS = {
    # name: kind, source, destination dims (C order), entries/row, fields
    'config1': dict(kind='bilinear', src_dims=(180, 360),
                    dst_dims=(360, 720), K=1,
                    title='1deg -> 0.5deg lat-lon bilinear, one 2-D field'),
    # config 1 as ESMF makes it (pyremap_amd.weights.bilinear_3d reproduces
    # ESMF's weights): quads of source centres along 3-D lines, and -- what
    # matters to the kernels -- the pole caps: the 2 x 720 destination cells
    # beyond the last source row take that whole row, 360 entries each, a
    # third of all entries
    'config1_esmf': dict(kind='esmf_bilinear', src_res=1.0, dst_res=0.5,
                         dst_dims=(360, 720), K=1, seed=11,
                         title='1deg -> 0.5deg lat-lon bilinear as ESMF '
                               'makes it (pole caps), one 2-D field'),
    'config2': dict(kind='conservative', n_a=7153, dst_dims=(180, 360),
                    k_lo=1, k_hi=4, K=64, empty_frac=0.3,
                    title='QU240 -> 1deg conservative, 64 fields'),
    'config3': dict(kind='conservative', n_a=235160, dst_dims=(360, 720),
                    k_lo=3, k_hi=7, K=512, empty_frac=0.3,
                    title='EC30to60 -> 0.5deg conservative, 512 fields'),
    'config4': dict(kind='bilinear', src_dims=(834, 1001),
                    dst_dims=(5001, 6001), K=128,
                    title='6 km -> 1 km Antarctic stereographic bilinear, '
                          '128 fields'),
    'config5': dict(kind='conservative', n_a=3693225, dst_dims=(1800, 3600),
                    k_lo=12, k_hi=30, K=1024, empty_frac=0.3, signed=True,
                    title='oRRS18to6 -> 0.1deg 2nd-order conservative, '
                          '1024 fields'),
    'headline': dict(kind='conservative', n_a=3693225, dst_dims=(720, 1440),
                     k_lo=4, k_hi=12, K=512, empty_frac=0.3,
                     title='oRRS18to6 (3.7 M) -> 0.25deg (1.0 M) '
                           'conservative, 512 fields'),
}


def _torch():
    import torch
    return torch


def _gen(seed, device):
    torch = _torch()
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


class SyntheticMap:
    """Triplets as a mapping file stores them, plus the sizes."""

    def __init__(self, row, col, S, frac_b, n_a, n_b, src_dims, dst_dims):
        self.row = row            # int32, 1-based, unsorted
        self.col = col            # int32, 1-based
        self.S = S                # float64
        self.frac_b = frac_b      # float64[n_b]
        self.n_a = int(n_a)
        self.n_b = int(n_b)
        self.src_dims = tuple(int(d) for d in src_dims)   # C order
        self.dst_dims = tuple(int(d) for d in dst_dims)   # C order

    @property
    def n_s(self):
        return int(self.S.shape[0])

    def numpy(self):
        import numpy as np
        return dict(
            row=self.row.cpu().numpy(), col=self.col.cpu().numpy(),
            S=self.S.cpu().numpy(), frac_b=self.frac_b.cpu().numpy(),
            n_a=np.int64(self.n_a), n_b=np.int64(self.n_b),
            src_grid_dims=np.asarray(self.src_dims[::-1], dtype=np.int32),
            dst_grid_dims=np.asarray(self.dst_dims[::-1], dtype=np.int32))

    def save(self, filename):
        from pyremap_amd.io.mapfile import write_mapping
        m = self.numpy()
        write_mapping(filename, m['n_a'], m['n_b'], m['src_grid_dims'],
                      m['dst_grid_dims'], m['row'], m['col'], m['S'],
                      m['frac_b'])


def _shuffle(row, col, S, gen):
    """Mapping files are not row-sorted (multi-PET ESMF output)."""
    torch = _torch()
    perm = torch.randperm(row.shape[0], generator=gen, device=row.device)
    return row[perm], col[perm], S[perm]


def bilinear_map(src_dims, dst_dims, seed=0, device='cpu', shuffle=True):
    """Tensor-product bilinear interpolation from a regular (ny, nx) source
    grid to a regular (my, mx) destination grid covering the same box."""
    torch = _torch()
    ny, nx = src_dims
    my, mx = dst_dims
    dev = torch.device(device)

    def axis(n_src, n_dst):
        # destination point j sits at t in [0, n_src - 1]
        t = torch.arange(n_dst, device=dev, dtype=torch.float64) * \
            ((n_src - 1) / max(n_dst - 1, 1))
        i0 = torch.clamp(t.floor().to(torch.int64), 0, max(n_src - 2, 0))
        w1 = t - i0.to(torch.float64)
        return i0, 1.0 - w1, w1

    y0, wy0, wy1 = axis(ny, my)
    x0, wx0, wx1 = axis(nx, mx)
    rows = torch.arange(my * mx, device=dev, dtype=torch.int64)
    jy = rows // mx
    jx = rows - jy * mx
    cols, vals = [], []
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            yy = torch.clamp(y0[jy] + dy, max=ny - 1)
            xx = torch.clamp(x0[jx] + dx, max=nx - 1)
            cols.append(yy * nx + xx)
            vals.append(wy[jy] * wx[jx])
    row = rows.repeat(4)
    col = torch.cat(cols)
    S = torch.cat(vals)
    keep = S != 0.0          # edge points have exact zero weights: drop them
    row, col, S = row[keep], col[keep], S[keep]
    if shuffle:
        row, col, S = _shuffle(row, col, S, _gen(seed, dev))
    frac_b = torch.ones(my * mx, device=dev, dtype=torch.float64)
    return SyntheticMap((row + 1).to(torch.int32), (col + 1).to(torch.int32),
                        S, frac_b, ny * nx, my * mx, (ny, nx), (my, mx))


def conservative_map(n_a, dst_dims, k_lo, k_hi, seed=0, device='cpu',
                     empty_frac=0.3, signed=False, locality='raster',
                     shuffle=True, cover=True):
    """Overlap-like weights from an unstructured n_a-cell mesh to a regular
    destination grid (see the module docstring)."""
    torch = _torch()
    dev = torch.device(device)
    gen = _gen(seed, dev)
    my, mx = dst_dims
    n_b = my * mx
    # neighbourhood (in destination cells) large enough for k_hi draws
    half_y, half_x = 1, 1
    while (2 * half_y + 1) * (2 * half_x + 1) < k_hi:
        if half_x <= half_y:
            half_x += 1
        else:
            half_y += 1
    win_x = 2 * half_x + 1
    win = (2 * half_y + 1) * win_x

    rows = torch.arange(n_b, device=dev, dtype=torch.int64)
    jy = rows // mx
    jx = rows - jy * mx
    # blocky land mask: coarse random field, upsampled
    by, bx = max(my // 15, 1), max(mx // 15, 1)
    coarse = torch.rand((by, bx), generator=gen, device=dev) < empty_frac
    land = coarse[(jy * by) // my, (jx * bx) // mx]
    ocean = ~land
    n_ocean = max(int(ocean.sum()), 1)
    # The source mesh exists only where there is ocean: its cells are
    # numbered along the row-major order of the ocean destination cells,
    # n_a / n_ocean source cells per destination cell.
    rank = torch.cumsum(ocean.to(torch.int64), 0) - 1
    ratio = n_a / n_ocean

    k = torch.randint(k_lo, k_hi + 1, (n_b,), generator=gen, device=dev)
    k = torch.where(land, torch.zeros_like(k), k)

    # a random order of the window per row; the first k entries are taken
    chunks_row, chunks_col, chunks_val = [], [], []
    step = max(1, (1 << 24) // win)      # bound temporary memory
    for a in range(0, n_b, step):
        b = min(a + step, n_b)
        n = b - a
        order = torch.rand((n, win), generator=gen, device=dev).argsort(dim=1)
        take = torch.arange(win, device=dev).unsqueeze(0) < \
            k[a:b].unsqueeze(1)
        sel = order[take]                      # window slots, row-major
        r = rows[a:b].unsqueeze(1).expand(n, win)[take]
        if locality in ('raster', 'mesh', 'scatter'):
            oy = sel // win_x - half_y
            ox = sel % win_x - half_x
            qy = torch.clamp(jy[r] + oy, 0, my - 1)
            qx = (jx[r] + ox) % mx
            q = qy * mx + qx
            q = torch.where(ocean[q], q, r)    # land neighbour: own cell
            u = torch.rand(sel.shape[0], generator=gen, device=dev,
                           dtype=torch.float64)
            c = ((rank[q].to(torch.float64) + u) * ratio).to(torch.int64)
            c = torch.clamp(c, 0, n_a - 1)
        elif locality == 'none':
            # same counts, source cells anywhere: worst-case gather
            c = torch.randint(0, n_a, (sel.shape[0],), generator=gen,
                              device=dev)
        else:
            raise ValueError(f'unknown locality {locality!r}')
        w = torch.rand(sel.shape[0], generator=gen, device=dev,
                       dtype=torch.float64) + 0.05
        if signed:
            # 2nd-order conservative: some negative gradient terms
            neg = torch.rand(sel.shape[0], generator=gen, device=dev) < 0.25
            w = torch.where(neg, -0.3 * w, w)
        chunks_row.append(r)
        chunks_col.append(c)
        chunks_val.append(w)
    row = torch.cat(chunks_row) if chunks_row else rows[:0]
    col = torch.cat(chunks_col) if chunks_col else rows[:0]
    S = torch.cat(chunks_val) if chunks_val else \
        torch.zeros(0, device=dev, dtype=torch.float64)

    if cover and locality != 'none' and row.numel():
        # A conservative map touches EVERY source cell (each one overlaps at
        # least the destination cell it lies in).  The random draws above
        # miss some (1 % on config 3, 20 % on the headline, whose source is
        # 5 x finer than its destination): give each missed source cell to
        # the ocean destination cell that owns it in the raster numbering.
        hit = torch.zeros(n_a, dtype=torch.bool, device=dev)
        hit[col] = True
        orphan = (~hit).nonzero().squeeze(1)
        if orphan.numel():
            ocean_rows = ocean.nonzero().squeeze(1)
            owner = torch.clamp((orphan.to(torch.float64) / ratio)
                                .to(torch.int64), 0, n_ocean - 1)
            w = torch.rand(orphan.shape[0], generator=gen, device=dev,
                           dtype=torch.float64) + 0.05
            row = torch.cat([row, ocean_rows[owner]])
            col = torch.cat([col, orphan])
            S = torch.cat([S, w])

    if locality == 'mesh' and int(ocean.sum()) > 0:
        # same overlaps, the source cells numbered as a mesh generator
        # numbers them: a column permutation of the 'raster' matrix
        ocean_rows = ocean.nonzero().squeeze(1)
        owner = torch.clamp((torch.arange(n_a, device=dev,
                                          dtype=torch.float64) / ratio)
                            .to(torch.int64), 0, n_ocean - 1)
        sub = torch.arange(n_a, device=dev, dtype=torch.float64) / ratio - \
            owner.to(torch.float64)
        q = ocean_rows[owner]
        col = mesh_numbering(jy[q].to(torch.float64),
                             jx[q].to(torch.float64) + sub, seed=seed)[col]

    if locality == 'scatter':
        # same overlaps, source ids at random: no id locality whatsoever
        col = torch.randperm(n_a, generator=gen, device=dev)[col]

    # scale rows to sum to frac_b in (0, 1]
    rowsum = torch.zeros(n_b, device=dev, dtype=torch.float64)
    rowsum.index_add_(0, row, S)
    frac_b = torch.rand(n_b, generator=gen, device=dev,
                        dtype=torch.float64) * 0.9 + 0.1
    frac_b = torch.where(k > 0, frac_b, torch.zeros_like(frac_b))
    safe = torch.where(rowsum.abs() > 1e-3, rowsum, torch.ones_like(rowsum))
    S = S * (frac_b / safe)[row]
    if shuffle:
        row, col, S = _shuffle(row, col, S, gen)
    return SyntheticMap((row + 1).to(torch.int32), (col + 1).to(torch.int32),
                        S, frac_b, n_a, n_b, (n_a,), (my, mx))


def mesh_numbering(py, px, seed=0, scramble=True):
    """
    Number cells the way an MPAS mesh does, given their positions (any two
    coordinates in which neighbours are close): returns the int64 tensor
    ``new_id[cell]``, a permutation of ``range(len(py))``.

    MPAS meshes made by icosahedral bisection (the reference's QU240 fixture,
    ``tests/test_interpolate/mpasMesh.nc``) number the coarsest points first
    and every refinement level behind the previous ones, the new points of a
    level in the order of the (coarser) points they were created around.
    Measured on that file: a cell's id neighbours are its spatial neighbours
    (the rings of 6 around a parent: median distance between consecutive ids
    1.7 cell spacings), but of a cell's spatial neighbours only 20 % have an
    id within 2 of its own and 75 % are more than n/6 away (median n/2.8); the
    ids met along one 1-degree latitude row span 90 % of the id range.

    The same construction on a quad hierarchy: cells are ranked along a
    Morton curve through their positions (the top of the hierarchy scrambled,
    as the icosahedron's faces come in no raster order); the base-4 digits of
    the rank are the path through the hierarchy; a cell whose lowest non-zero
    digit is digit t belongs to level t (3/4 of the cells to the finest), its
    parent is the cell with that digit cleared, and level t is numbered
    behind all coarser levels, its cells sorted by their parent's NEW id.
    """
    torch = _torch()
    dev = py.device
    n = int(py.shape[0])
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    py = py.to(torch.float64)
    px = px.to(torch.float64)
    ey = float(py.max() - py.min()) + 1e-9
    ex = float(px.max() - px.min()) + 1e-9
    spacing = max((ey * ex / n) ** 0.5, 1e-12)
    scale = min(4.0 / spacing, ((1 << 20) - 1) / max(ey, ex))
    qy = ((py - py.min()) * scale).to(torch.int64)
    qx = ((px - px.min()) * scale).to(torch.int64)
    code = torch.zeros(n, dtype=torch.int64, device=dev)
    for bit in range(20):
        code |= ((qx >> bit) & 1) << (2 * bit)
        code |= ((qy >> bit) & 1) << (2 * bit + 1)
    # rank along the curve (ties: input order)
    by_code = torch.argsort(code, stable=True)
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    rank[by_code] = torch.arange(n, device=dev)
    digits = 1
    while 4 ** digits < n:
        digits += 1
    if scramble and digits >= 3:
        bs = 4 ** (digits - 2)
        whole = n // bs
        g = torch.Generator(device='cpu')
        g.manual_seed(int(seed) + 7919)
        perm = torch.arange((n + bs - 1) // bs, dtype=torch.int64)
        perm[:whole] = torch.randperm(whole, generator=g)
        rank = perm.to(dev)[rank // bs] * bs + rank % bs
    # trailing zero base-4 digits = level (0: finest); rank 0 is the root
    level = torch.full((n,), digits, dtype=torch.int64, device=dev)
    for t in range(digits - 1, -1, -1):
        level = torch.where((rank >> (2 * t)) & 3 != 0,
                            torch.full_like(level, t), level)
    cell_of_rank = torch.empty(n, dtype=torch.int64, device=dev)
    cell_of_rank[rank] = torch.arange(n, device=dev)
    new_of_rank = torch.zeros(n, dtype=torch.int64, device=dev)
    done = 1                                  # the root keeps id 0
    for t in range(digits - 1, -1, -1):
        r = (level[cell_of_rank] == t).nonzero().squeeze(1)   # ranks, level t
        if r.numel() == 0:
            continue
        d = (r >> (2 * t)) & 3
        parent = r & ~(3 << (2 * t))
        key = new_of_rank[parent] * 4 + d
        new_of_rank[r[torch.argsort(key, stable=True)]] = \
            done + torch.arange(r.numel(), device=dev)
        done += int(r.numel())
    return new_of_rank[rank]


def numbering_stats(py, px, ids=None, row_of=None):
    """
    The statistics :func:`mesh_numbering` is calibrated with, for cells at
    positions ``(py, px)`` carrying ``ids`` (default: their index):
    median distance between consecutive ids in mean cell spacings, the id
    jumps between each cell and its 4 nearest neighbours (fraction within 2,
    median / n), and -- with ``row_of`` (an integer per cell: the destination
    grid row it lies in) -- the median fraction of the id range one such row
    spans.  Brute force in chunks: for tests and DESIGN.md, not for the path.
    """
    torch = _torch()
    n = int(py.shape[0])
    ids = torch.arange(n) if ids is None else ids.cpu()
    py, px = py.cpu().to(torch.float64), px.cpu().to(torch.float64)
    pos = torch.stack([py, px], 1)
    by_id = torch.argsort(ids)
    p = pos[by_id]
    ey = float(py.max() - py.min()) + 1e-9
    ex = float(px.max() - px.min()) + 1e-9
    spacing = (ey * ex / n) ** 0.5
    step = (p[1:] - p[:-1]).norm(dim=1) / spacing
    g = torch.Generator()
    g.manual_seed(0)
    sample = torch.randperm(n, generator=g)[:2000]
    jumps = []
    for a in range(0, sample.numel(), 250):
        sidx = sample[a:a + 250]
        d = torch.cdist(pos[sidx], pos)
        nn = d.topk(5, largest=False).indices[:, 1:]
        jumps.append((ids[nn] - ids[sidx].unsqueeze(1)).abs().flatten())
    jumps = torch.cat(jumps).to(torch.float64)
    out = dict(n=n, consecutive_id_distance_median=float(step.median()),
               neighbour_jump_within_2=float((jumps <= 2).double().mean()),
               neighbour_jump_median_over_n=float(jumps.median()) / n,
               neighbour_jump_q25_over_n=float(jumps.quantile(0.25)) / n)
    if row_of is not None:
        row_of = row_of.cpu()
        spans = []
        for r in torch.unique(row_of).tolist():
            sel = ids[row_of == r]
            if sel.numel() >= 8:
                spans.append(float(sel.max() - sel.min()) / n)
        if spans:
            out['row_id_span_median'] = float(
                torch.tensor(spans).median())
    return out


def knn_map(lat_src, lon_src, dst_dims, k_hi=4, seed=0, device='cpu',
            reach=1.2, shuffle=True):
    """
    Overlap-like weights from REAL cell centres (radians; e.g. ``latCell`` /
    ``lonCell`` of the QU240 mesh, ``tests/golden/qu240_cells.npz``) to a
    regular ``dst_dims = (nlat, nlon)`` global grid: every destination cell
    takes the (at most ``k_hi``) source cells whose centres lie within
    ``reach`` source spacings of its own, weights falling off linearly with
    distance, rows scaled to ``frac_b``; destination cells with no source
    cell in reach (land on an ocean mesh) stay empty with ``frac_b = 0``.
    The column indices are the mesh's own cell numbers -- what a mapping file
    made from that mesh holds.
    """
    torch = _torch()
    dev = torch.device(device)
    gen = _gen(seed, dev)
    lat = torch.as_tensor(lat_src, dtype=torch.float64, device=dev)
    lon = torch.as_tensor(lon_src, dtype=torch.float64, device=dev)
    n_a = int(lat.shape[0])
    my, mx = (int(d) for d in dst_dims)
    n_b = my * mx
    src = torch.stack([lat.cos() * lon.cos(), lat.cos() * lon.sin(),
                       lat.sin()], 1)
    # mean spacing of the source cells where there are any (chord length)
    d_nn = []
    for a in range(0, min(n_a, 2000), 500):
        d = torch.cdist(src[a:a + 500], src)
        d_nn.append(d.topk(2, largest=False).values[:, 1])
    spacing = float(torch.cat(d_nn).median())
    dlat = (torch.arange(my, device=dev, dtype=torch.float64) + 0.5) * \
        (torch.pi / my) - torch.pi / 2
    dlon = (torch.arange(mx, device=dev, dtype=torch.float64) + 0.5) * \
        (2 * torch.pi / mx) - torch.pi
    rows, cols, vals = [], [], []
    per = max(1, (1 << 25) // max(n_a, 1))
    for a in range(0, n_b, per):
        r = torch.arange(a, min(a + per, n_b), device=dev)
        la, lo = dlat[r // mx], dlon[r % mx]
        dst = torch.stack([la.cos() * lo.cos(), la.cos() * lo.sin(),
                           la.sin()], 1)
        d, idx = torch.cdist(dst, src).topk(k_hi, largest=False)
        w = 1.0 - d / (reach * spacing)
        keep = w > 0
        rows.append(r.unsqueeze(1).expand_as(idx)[keep])
        cols.append(idx[keep])
        vals.append(w[keep] + 0.05)
    row, col, S = torch.cat(rows), torch.cat(cols), torch.cat(vals)
    rowsum = torch.zeros(n_b, device=dev, dtype=torch.float64)
    rowsum.index_add_(0, row, S)
    frac_b = torch.rand(n_b, generator=gen, device=dev,
                        dtype=torch.float64) * 0.9 + 0.1
    frac_b = torch.where(rowsum > 0, frac_b, torch.zeros_like(frac_b))
    S = S * (frac_b / torch.where(rowsum > 0, rowsum,
                                  torch.ones_like(rowsum)))[row]
    if shuffle:
        row, col, S = _shuffle(row, col, S, gen)
    return SyntheticMap((row + 1).to(torch.int32), (col + 1).to(torch.int32),
                        S, frac_b, n_a, n_b, (n_a,), (my, mx))


def make_config(name, device='cpu', seed=None, locality='raster'):
    """The synthetic mapping of one of :data:`CONFIGS`."""
    cfg = CONFIGS[name]
    if seed is None:
        # (configs added later carry their seed: the others keep theirs)
        seed = cfg['seed'] if 'seed' in cfg else sorted(
            k for k in CONFIGS if 'seed' not in CONFIGS[k]).index(name)
    if cfg['kind'] == 'esmf_bilinear':
        from pyremap_amd.descriptor import get_lat_lon_descriptor
        from pyremap_amd.weights import build_weights
        torch = _torch()
        m = build_weights(
            get_lat_lon_descriptor(cfg['src_res'], cfg['src_res']),
            get_lat_lon_descriptor(cfg['dst_res'], cfg['dst_res']),
            'bilinear')
        row, col, S = _shuffle(torch.as_tensor(m.row, device=device),
                               torch.as_tensor(m.col, device=device),
                               torch.as_tensor(m.S, device=device),
                               _gen(seed, device))
        return SyntheticMap(row, col, S,
                            torch.as_tensor(m.frac_b, device=device),
                            m.n_a, m.n_b, m.src_grid_dims[::-1],
                            m.dst_grid_dims[::-1])
    if cfg['kind'] == 'bilinear':
        return bilinear_map(cfg['src_dims'], cfg['dst_dims'], seed=seed,
                            device=device)
    return conservative_map(
        cfg['n_a'], cfg['dst_dims'], cfg['k_lo'], cfg['k_hi'], seed=seed,
        device=device, empty_frac=cfg.get('empty_frac', 0.3),
        signed=cfg.get('signed', False), locality=locality)
