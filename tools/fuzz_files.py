#!/usr/bin/env python3
"""
Time-bounded fuzz of the file -> file path (GPU box only): random Datasets
(float / int / char variables with and without the source dims, dims in
random order, NaNs, a record dimension or none) written in a random format,
remapped by `Remapper.ncremap` -- all at once and STREAMED (every variable
above a 1-byte threshold: lazy reads, header written last) -- and compared
with `remap_numpy` of the same Dataset, value for value, attribute for
attribute.

    python tools/fuzz_files.py [seconds=300] [first_seed=0]
"""
import os
import shutil
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pyremap_amd import DataArray, Dataset, Remapper, synthetic  # noqa: E402
from pyremap_amd.io.netcdf import open_dataset, write_netcdf  # noqa: E402
from pyremap_amd.remapper import remap_file  # noqa: E402

FORMATS = ['NETCDF3_CLASSIC', 'NETCDF3_64BIT', 'NETCDF3_64BIT_DATA',
           'NETCDF4']


def one(seed, tmp):
    rng = np.random.default_rng(55_000 + seed)
    n_a = int(rng.integers(300, 3000))
    dst = (int(rng.integers(6, 40)), int(rng.integers(6, 40)))
    m = synthetic.conservative_map(n_a, dst, 1, 6, seed=seed,
                                   locality=str(rng.choice(['raster',
                                                            'mesh'])))
    map_path = os.path.join(tmp, f'map_{seed}.nc')
    m.save(map_path)

    class Desc:
        pass
    s, d = Desc(), Desc()
    s.dims, s.dim_sizes = ['nCells'], [n_a]
    d.dims, d.dim_sizes = ['lat', 'lon'], list(dst)
    d.coords, d.mesh_name = {}, 'fuzz'
    sizes = {'Time': int(rng.integers(1, 5)), 'nCells': n_a,
             'z': int(rng.integers(1, 9)), 'w': int(rng.integers(1, 4)),
             'StrLen': 8}
    ds = Dataset(attrs={'title': f'seed {seed}'})
    for v in range(int(rng.integers(1, 7))):
        # (at least one variable on the source mesh: without any the
        # reference raises KeyError at `ds.sizes[dim]`, remap_numpy.py:33-38,
        # and so does this build)
        kind = 'src_f8' if v == 0 else rng.choice(
            ['src_f8', 'src_f4', 'src_i4', 'plain_f8', 'plain_i4', 'char'])
        if kind.startswith('src'):
            extra = [x for x in ('Time', 'z', 'w') if rng.random() < 0.5]
            dims = extra[:]
            dims.insert(int(rng.integers(0, len(dims) + 1)), 'nCells')
            if 'Time' in dims:          # the record dimension leads
                dims.remove('Time')
                dims.insert(0, 'Time')
        elif kind == 'char':
            dims = ['Time', 'StrLen']
        else:
            dims = [x for x in ('Time', 'z') if rng.random() < 0.6] or ['z']
        shape = [sizes[x] for x in dims]
        if kind == 'char':
            data = rng.integers(97, 123, size=shape).astype(np.uint8) \
                .view('S1')
        elif kind.endswith('i4'):
            data = rng.integers(-99, 99, size=shape).astype(np.int32)
        else:
            data = rng.standard_normal(shape).astype(
                np.float32 if kind.endswith('f4') else np.float64)
            if rng.random() < 0.5:
                data[rng.random(shape) < 0.15] = np.nan
        ds[f'v{v}_{kind}'] = DataArray(data, dims=dims,
                                       attrs={'long_name': f'v{v}'})
    fmt = str(rng.choice(FORMATS))
    unlimited = ['Time'] if rng.random() < 0.6 and any(
        'Time' in ds[v].dims for v in ds.data_vars) else []
    in_path = os.path.join(tmp, f'in_{seed}.nc')
    write_netcdf(ds, in_path, format=fmt, unlimited_dims=unlimited)
    thr = None if rng.random() < 0.4 else float(rng.choice([0.0, 0.2]))
    variables = None
    if rng.random() < 0.3:
        names = list(ds.data_vars)
        variables = [names[0]] + [n for n in names[1:]
                                  if rng.random() < 0.6]
    r = Remapper(map_filename=map_path, src_descriptor=s, dst_descriptor=d)
    src_ds = open_dataset(in_path, variables=variables)
    if variables is not None:
        src_ds = src_ds.drop_vars([v for v in src_ds.data_vars
                                   if v not in variables])
    ref = r.remap_numpy(src_ds, thr)
    what = f'seed {seed} {fmt} unlimited {unlimited} thr {thr} vars {variables}'
    outs = {}
    for tag, stream_bytes in (('eager', 1 << 40), ('streamed', 1)):
        remap_file.STREAM_BYTES = stream_bytes
        out_path = os.path.join(tmp, f'out_{seed}_{tag}.nc')
        r.ncremap(in_path, out_path, variable_list=variables,
                  renormalize=thr, overwrite=True)
        outs[tag] = open_dataset(out_path, mask_and_scale=False)
        back = open_dataset(out_path)
        assert list(back.data_vars) == list(ref.data_vars), what
        for name in ref.data_vars:
            a, b = ref[name], back[name]
            assert a.dims == b.dims, (what, name, a.dims, b.dims)
            if a.values.dtype.kind == 'f':
                assert np.array_equal(
                    np.asarray(a.values, dtype=np.float64),
                    np.asarray(b.values, dtype=np.float64),
                    equal_nan=True), (what, name, tag)
            else:
                assert a.values.tobytes() == b.values.tobytes(), \
                    (what, name, tag)
    a, b = outs['eager'], outs['streamed']
    for name in a.variables:
        va, vb = a.variables[name], b.variables[name]
        assert va.dims == vb.dims and va.dtype == vb.dtype, (what, name)
        assert sorted(va.attrs) == sorted(vb.attrs), (what, name)
        assert va.values.tobytes() == vb.values.tobytes(), (what, name)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    assert torch.cuda.is_available()
    tmp = tempfile.mkdtemp()
    keep = remap_file.STREAM_BYTES
    t0, n, bad = time.time(), 0, []
    try:
        while time.time() - t0 < budget and len(bad) < 4:
            try:
                one(seed, tmp)
                n += 1
            except Exception as exc:   # noqa: BLE001 - reported
                print('SEED', seed, 'FAILED', type(exc).__name__,
                      str(exc)[:400])
                bad.append(seed)
            seed += 1
    finally:
        remap_file.STREAM_BYTES = keep
        shutil.rmtree(tmp, ignore_errors=True)
    print(f'seeds ok: {n}, failed: {bad}, next seed {seed}, '
          f'{time.time() - t0:.0f} s')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
