#!/usr/bin/env python3
"""
How ESMF cuts the polygons of a dual mesh into triangles before it
interpolates (bilinear, MPAS vertices as the source): inferred from the output
the reference's tests store for `test_mpas_vertex_to_latlon`
(tests/golden/ref_fixtures/ref_mpas_vertex_to_latlon.nc, ESMF weights).

For every cell of the QU240 mesh (a polygon of 5-7 vertices) every
triangulation of the polygon (14 for a hexagon) is tried on the destination
points inside it: a triangulation is CONSISTENT when every such point,
interpolated linearly in the triangle that holds it (straight lines in 3-D),
reproduces the stored value.  Then greedy ear-clipping rules are scored
against the consistent triangulation of every cell that has exactly one.

    python tools/esmf_rule_probe.py        (CPU only, about two minutes)
"""
import collections
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from pyremap_amd.io.netcdf import open_dataset  # noqa: E402
from pyremap_amd.weights import _unit, clip_ears  # noqa: E402


def triangulations(poly):
    if len(poly) == 3:
        return [[tuple(poly)]]
    out = []
    a, b = poly[0], poly[-1]
    for k in range(1, len(poly) - 1):
        left = triangulations(poly[:k + 1]) if k >= 2 else [[]]
        right = triangulations(poly[k:]) if len(poly) - k >= 3 else [[]]
        out += [lf + [(a, poly[k], b)] + rt for lf in left for rt in right]
    return out


def bary(a, b, c, q):
    w = np.linalg.solve(np.stack([a, b, c], -1), q)
    return None if w.sum() <= 0 else w / w.sum()


def main():
    gold = os.path.join(REPO, 'tests', 'golden')
    mesh = open_dataset(os.path.join(gold, 'ref_fixtures', 'mpasMesh.nc'))
    pc = _unit(mesh['latCell'].values, mesh['lonCell'].values)
    pv = _unit(mesh['latVertex'].values, mesh['lonVertex'].values)
    voc = mesh['verticesOnCell'].values - 1
    ne = mesh['nEdgesOnCell'].values
    f = open_dataset(os.path.join(
        gold, 'hdf5', 'nc4_mpasAreaVertex.nc'))['areaVertex'].values
    ref = open_dataset(os.path.join(gold, 'ref_fixtures',
                                    'ref_mpas_vertex_to_latlon.nc'))
    want = ref['areaVertex'].values.ravel()
    lat, lon = np.meshgrid(np.deg2rad(ref['lat'].values),
                           np.deg2rad(ref['lon'].values), indexing='ij')
    q_all = _unit(lat.ravel(), lon.ravel())
    # destination points by the cell that holds them (nearest three centres)
    near = np.argsort(-(q_all @ pc.T), axis=1)[:, :3] \
        if len(q_all) * len(pc) < 5e8 else None
    if near is None:
        near = np.stack([np.argsort(-(q_all[i:i + 2000] @ pc.T),
                                    axis=1)[:, :3]
                         for i in range(0, len(q_all), 2000)])
        near = near.reshape(-1, 3)
    points = collections.defaultdict(list)
    for qi in np.nonzero(~np.isnan(want))[0]:
        for c in near[qi]:
            n, vs = ne[c], voc[c, :ne[c]]
            if any((w := bary(pc[c], pv[vs[i]], pv[vs[(i + 1) % n]],
                              q_all[qi])) is not None and
                   (w >= -1e-12).all() for i in range(n)):
                points[c].append(qi)
                break
    tri_of = {n: triangulations(tuple(range(n))) for n in (5, 6, 7)}
    chosen = {}
    for c, qis in points.items():
        n, vs = ne[c], voc[c, :ne[c]]
        ok = []
        for ti, tris in enumerate(tri_of[n]):
            good = True
            for qi in qis:
                hit = False
                for t in tris:
                    o = [vs[t[0]], vs[t[1]], vs[t[2]]]
                    w = bary(pv[o[0]], pv[o[1]], pv[o[2]], q_all[qi])
                    if w is not None and (w >= -1e-10).all():
                        hit = abs(w @ f[o] - want[qi]) < 1e-9 * abs(want[qi])
                        break
                if not hit:
                    good = False
                    break
            if good:
                ok.append(ti)
        chosen[c] = ok
    print('cells by (corners, consistent triangulations):',
          dict(collections.Counter((int(ne[c]), len(v))
                                   for c, v in chosen.items())))
    unique = {c: v[0] for c, v in chosen.items() if len(v) == 1}

    def measures(p, i, j, k):
        a, b, c = p[i], p[j], p[k]

        def ang(u, v):
            return np.arccos(np.clip(u @ v / np.linalg.norm(u) /
                                     np.linalg.norm(v), -1, 1))
        return dict(angle=ang(a - b, c - b),
                    area=np.linalg.norm(np.cross(b - a, c - a)),
                    diagonal=np.linalg.norm(c - a), dot=(a - b) @ (c - b))

    def greedy(p, n, key, sign):
        left, tris = list(range(n)), []
        while len(left) > 3:
            m = len(left)
            x = max(range(m), key=lambda x: sign * measures(
                p, left[x - 1], left[x], left[(x + 1) % m])[key])
            tris.append(frozenset((left[x - 1], left[x], left[(x + 1) % m])))
            del left[x]
        return frozenset(tris + [frozenset(left)])
    score = collections.Counter()
    for c, ti in unique.items():
        n = ne[c]
        p = pv[voc[c, :n]]
        target = frozenset(frozenset(t) for t in tri_of[n][ti])
        for key in ('angle', 'area', 'diagonal', 'dot'):
            for sign, word in ((1, 'largest'), (-1, 'smallest')):
                score[f'{word} {key}'] += greedy(p, n, key, sign) == target
        fan = frozenset(frozenset((0, i, i + 1)) for i in range(1, n - 1))
        score['fan from the first corner'] += fan == target
        mine = clip_ears(p, np.arange(n)[None, :], np.array([n]))
        score['weights.clip_ears'] += frozenset(
            frozenset(int(v) for v in t) for t in mine) == target
    print(f'{len(unique)} cells with exactly one consistent triangulation; '
          f'cells each rule reproduces:')
    for rule, hits in score.most_common():
        print(f'   {rule:28s} {hits}')


if __name__ == '__main__':
    main()
