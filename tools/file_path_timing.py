#!/usr/bin/env python3
"""
Where the time of a file -> file remap goes (GPU box only): the reference's
`Remapper.ncremap()` call (`pyremap/remapper/ncremap.py:15-145`) on the metric
configuration's shapes -- an MPAS-like `(Time, nCells, nVertLevels)` file in,
a `(Time, lat, lon, nVertLevels)` file out.

    python tools/file_path_timing.py [--fields 512] [--format NETCDF4]

Prints one JSON line: seconds to write the synthetic inputs (not part of the
path), to load the mapping file into a device plan, and for `ncremap` --
split into reading the input, remapping (PCIe both ways + kernel) and
writing the output -- beside the kernel's own time.
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='config3')
    ap.add_argument('--fields', type=int, default=512)
    ap.add_argument('--levels', type=int, default=64)
    ap.add_argument('--format', default='NETCDF3_64BIT_DATA')
    ap.add_argument('--dir', default=None)
    ap.add_argument('--renormalize', type=float, default=None)
    args = ap.parse_args()
    import torch
    from pyremap_amd import (
        DataArray,
        Dataset,
        LatLonGridDescriptor,
        MpasCellMeshDescriptor,
        Remapper,
        synthetic,
    )
    from pyremap_amd.io import netcdf
    from pyremap_amd.remapper import remap_file

    tmp = tempfile.mkdtemp(dir=args.dir)
    m = synthetic.make_config(args.workload)
    nlat, nlon = m.dst_dims
    map_path = os.path.join(tmp, f'map_{args.workload}_aave.nc')
    t = time.perf_counter()
    m.save(map_path)
    t_map_write = time.perf_counter() - t
    rng = np.random.default_rng(3)
    src = MpasCellMeshDescriptor(mesh_name='synthetic',
                                 lat=rng.random(m.n_a), lon=rng.random(m.n_a))
    dst = LatLonGridDescriptor.create(np.linspace(-90, 90, nlat + 1),
                                      np.linspace(-180, 180, nlon + 1))
    n_t = max(1, args.fields // args.levels)
    field = rng.standard_normal((n_t, m.n_a, args.levels))
    if args.renormalize is not None:
        field[:, rng.random(m.n_a) < 0.25, args.levels // 2:] = np.nan
    ds = Dataset(attrs={'source': 'synthetic'})
    ds['temperature'] = DataArray(field, dims=('Time', 'nCells',
                                               'nVertLevels'))
    in_path = os.path.join(tmp, 'in.nc')
    out_path = os.path.join(tmp, 'out.nc')
    t = time.perf_counter()
    netcdf.write_netcdf(ds, in_path, format=args.format,
                        unlimited_dims=['Time'])
    t_in_write = time.perf_counter() - t
    del ds, field

    t = time.perf_counter()
    remapper = Remapper(map_filename=map_path, src_descriptor=src,
                        dst_descriptor=dst)
    plan = remapper.load_mapping()
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t

    # the path, with its three phases timed from outside
    spans = {}
    real_open, real_write = remap_file.open_dataset, remap_file.write_netcdf
    real_remap = remap_file._remap_numpy

    def timed(name, fn):
        def wrapper(*a, **k):
            t0 = time.perf_counter()
            out = fn(*a, **k)
            # (large variables are read when the writer reaches them:
            # 'read' is the header and the small variables, 'write' holds
            # read + remap + write of the streamed ones)
            spans[name] = spans.get(name, 0.0) + time.perf_counter() - t0
            return out
        return wrapper

    remap_file.open_dataset = timed('read', real_open)
    remap_file._remap_numpy = timed('remap', real_remap)
    remap_file.write_netcdf = timed('write', real_write)
    runs = []
    try:
        for rep in range(2):
            spans.clear()
            if os.path.exists(out_path):
                os.remove(out_path)
            t = time.perf_counter()
            remapper.ncremap(in_path, out_path, renormalize=args.renormalize)
            total = time.perf_counter() - t
            runs.append(dict(total_s=total, **{f'{k}_s': v
                                               for k, v in spans.items()}))
    finally:
        remap_file.open_dataset = real_open
        remap_file._remap_numpy = real_remap
        remap_file.write_netcdf = real_write

    # the kernel alone, fields resident
    from pyremap_amd import engine
    x = torch.randn((n_t, m.n_a, args.levels), device=plan.device,
                    dtype=torch.float64)
    mode = engine.MODE_FRACB
    engine.remap_tensor(plan, m.dst_dims, x, [1], mode)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        engine.remap_tensor(plan, m.dst_dims, x, [1], mode)
    b.record()
    torch.cuda.synchronize()
    print(json.dumps({
        'workload': args.workload, 'format': args.format,
        'fields': n_t * args.levels,
        'input_MB': os.path.getsize(in_path) / 1e6,
        'output_MB': os.path.getsize(out_path) / 1e6,
        'map_MB': os.path.getsize(map_path) / 1e6,
        'not_on_the_path': {'write_map_s': t_map_write,
                            'write_input_s': t_in_write},
        'load_mapping_s': t_plan,
        'ncremap_runs': runs,
        'kernel_ms': a.elapsed_time(b) / 20,
        'tmpdir': tmp}))
    for p in (in_path, out_path, map_path):
        os.remove(p)
    os.rmdir(tmp)


if __name__ == '__main__':
    main()
