#!/usr/bin/env python3
"""
File -> file on a file larger than one wants in memory: four
`(Time = 8, nCells, nVertLevels = 64)` variables on config 3's mapping
(3.85 GB in, 4.25 GB out), streamed (variables above
PYREMAP_AMD_STREAM_BYTES are read, remapped and written one at a time) and
all at once.  Prints wall time and the growth of the peak RSS across the
call, each mode in a fresh process.

    python tools/stream_timing.py [workdir]
"""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, resource, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from pyremap_amd import (DataArray, Dataset, LatLonGridDescriptor,
                         MpasCellMeshDescriptor, Remapper, synthetic)
from pyremap_amd.io.netcdf import write_netcdf
tmp, mode, fmt, nvars = sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
cfg = synthetic.CONFIGS['config3']
n, (nlat, nlon), T, L = cfg['n_a'], cfg['dst_dims'], 8, 64
map_path = os.path.join(tmp, 'map.nc')
src_path = os.path.join(tmp, f'big_in_{fmt}_{nvars}.nc')
rng = np.random.default_rng(1)
if not os.path.exists(map_path):
    synthetic.make_config('config3', locality='mesh').save(map_path)
if not os.path.exists(src_path):
    ds = Dataset()
    for v in range(nvars):
        x = rng.standard_normal((T, n, L))
        if v % 2:
            x[:, rng.random(n) < 0.2, L // 2:] = np.nan
        ds[f'var{v}'] = DataArray(x, dims=('Time', 'nCells', 'nVertLevels'))
    write_netcdf(ds, src_path, format=fmt, unlimited_dims=['Time'])
    del ds, x
src = MpasCellMeshDescriptor(mesh_name='EC30to60', lat=rng.random(n),
                             lon=rng.random(n))
dst = LatLonGridDescriptor.create(np.linspace(-90, 90, nlat + 1),
                                  np.linspace(-180, 180, nlon + 1))
r = Remapper(map_filename=map_path, src_descriptor=src, dst_descriptor=dst)
r.load_mapping()
r.remap_array(np.zeros((n, 40)), [0], 0.1)        # runtime + kernels loaded
out = os.path.join(tmp, f'big_out_{mode}_{fmt}.nc')
best = None
for rep in range(3):
    base = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = time.perf_counter()
    r.ncremap(src_path, out, renormalize=0.05, overwrite=True)
    dt = time.perf_counter() - t0
    peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    if rep == 0:
        grown = (peak - base) * 1024
    best = dt if best is None else min(best, dt)
print('RESULT', mode, fmt, grown, best, os.path.getsize(src_path),
      os.path.getsize(out))
'''


def main():
    tmp = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp()
    os.makedirs(tmp, exist_ok=True)
    script = os.path.join(tmp, 'stream_worker.py')
    with open(script, 'w') as f:
        f.write(WORKER)
    for fmt, nvars in (('NETCDF3_64BIT_DATA', 4), ('NETCDF4', 4),
                       ('NETCDF3_64BIT_DATA', 8)):
        for mode, threshold in (('eager', str(1 << 40)),
                                ('streamed', str(64 << 20))):
            env = dict(os.environ, PYREMAP_AMD_STREAM_BYTES=threshold)
            proc = subprocess.run(
                [sys.executable, script, REPO, tmp, mode, fmt, str(nvars)],
                capture_output=True, text=True, env=env)
            line = [ln for ln in proc.stdout.splitlines()
                    if ln.startswith('RESULT')]
            if not line:
                print(proc.stderr[-2000:])
                continue
            _, mode, fmt, grown, secs, n_in, n_out = line[-1].split()
            print(f'{fmt:20s} {mode:9s} peak RSS + {int(grown) / 1e9:5.2f} GB'
                  f'   {float(secs):6.3f} s per call   '
                  f'({int(n_in) / 1e9:.2f} GB in, {int(n_out) / 1e9:.2f} GB '
                  f'out)', flush=True)


if __name__ == '__main__':
    main()
