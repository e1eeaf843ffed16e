#!/usr/bin/env python3
"""
A Dataset of many 2-D fields -- the climatology files MPAS-Analysis remaps:
`(Time = 1, nCells)` variables by the dozen -- through
`Remapper.remap_numpy(ds, threshold)` on config 3's mapping: milliseconds per
call and per variable, and where the host time goes (cProfile, top entries).

    python tools/many_vars.py [n_vars=40] [--profile]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pyremap_amd import DataArray, Dataset, Remapper, synthetic  # noqa: E402


def main():
    n_vars = int(sys.argv[1]) if len(sys.argv) > 1 and \
        sys.argv[1].isdigit() else 40
    m = synthetic.make_config('config3', device='cuda', locality='mesh')

    class Desc:
        pass
    src, dst = Desc(), Desc()
    src.dims, src.dim_sizes = ['nCells'], [m.n_a]
    dst.dims, dst.dim_sizes = ['lat', 'lon'], list(m.dst_dims)
    dst.coords, dst.mesh_name = {}, 'bench'
    mm = m.numpy()
    r = Remapper.from_triplets(mm['row'], mm['col'], mm['S'], mm['frac_b'],
                               src, dst, device='cuda')
    rng = np.random.default_rng(0)
    ds = Dataset()
    for v in range(n_vars):
        x = rng.standard_normal((1, m.n_a))
        if v % 2:
            x[:, rng.random(m.n_a) < 0.2] = np.nan
        ds[f'v{v}'] = DataArray(x, dims=('Time', 'nCells'))
    for thr in (0.01, None):
        r.remap_numpy(ds, thr)
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = r.remap_numpy(ds, thr)
            times.append(time.perf_counter() - t0)
        t = min(times)
        print(f'threshold {thr}: {n_vars} variables {t * 1e3:.2f} ms, '
              f'{t / n_vars * 1e6:.0f} us per variable')
        del out
    if '--profile' in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            r.remap_numpy(ds, 0.01)
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(28)


if __name__ == '__main__' and '--files' not in sys.argv:
    main()


def file_path():
    """The same Dataset file -> file (`ncremap`)."""
    import tempfile
    from pyremap_amd import MpasCellMeshDescriptor, get_lat_lon_descriptor
    from pyremap_amd.io.netcdf import write_netcdf
    n_vars = 40
    m = synthetic.make_config('config3', device='cuda', locality='mesh')
    tmp = tempfile.mkdtemp()
    map_path = os.path.join(tmp, 'map.nc')
    m.save(map_path)
    rng = np.random.default_rng(0)
    ds = Dataset()
    for v in range(n_vars):
        x = rng.standard_normal((1, m.n_a))
        if v % 2:
            x[:, rng.random(m.n_a) < 0.2] = np.nan
        ds[f'v{v}'] = DataArray(x, dims=('Time', 'nCells'))
    src = MpasCellMeshDescriptor(mesh_name='ec', lat=rng.random(m.n_a),
                                 lon=rng.random(m.n_a))
    dst = get_lat_lon_descriptor(dlon=0.5, dlat=0.5)
    for fmt in ('NETCDF3_64BIT_DATA', 'NETCDF4'):
        in_path = os.path.join(tmp, f'in_{fmt}.nc')
        write_netcdf(ds, in_path, format=fmt, unlimited_dims=['Time'])
        r = Remapper(map_filename=map_path, src_descriptor=src,
                     dst_descriptor=dst)
        r.load_mapping()
        out_path = os.path.join(tmp, f'out_{fmt}.nc')
        times = []
        for _ in range(4):
            t0 = time.perf_counter()
            r.ncremap(in_path, out_path, renormalize=0.01, overwrite=True)
            times.append(time.perf_counter() - t0)
        print(f'{fmt}: {n_vars} variables file -> file '
              f'{min(times) * 1e3:.1f} ms ({os.path.getsize(in_path) / 1e6:.0f}'
              f' MB in, {os.path.getsize(out_path) / 1e6:.0f} MB out)')


if __name__ == '__main__' and '--files' in sys.argv:
    file_path()
