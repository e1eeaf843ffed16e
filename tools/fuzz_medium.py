#!/usr/bin/env python3
"""
Time-bounded fuzz at medium size (GPU box only): mappings of 5 000-80 000
source cells onto 100-300 x 100-400 grids -- hundreds of workgroups per
launch, every K tile rule (flat, whole batches per tile, one batch per
chunk, one or two elements per lane) -- as `(n, K)` and `(Time, n, levels)`
fields, float32 / float64, frac_b / masked, whole plan and a row shard, bit
for bit against the oracle.  Round 3: the source cells numbered along the
raster, as an MPAS mesh, or at random; `(Time, n)` and `(Time, n, 1..7)`
fields (the lanes-across-rows kernels); the shard also in its PACKED column
space on `X[ucols]`; every third seed through a 2- or 3-"device"
MultiDeviceRemap.  Round 4: half the seeds also through the opaque C plan
handle (remap_plan_create / _apply, with and without
remap_plan_prepare_short_runs) -- its own routing against the same oracle.
Round 5: masks of whole cells / of columns / both; every masked case also
with REMAP_FLAG_CELL_MASKS and through remap_tensor_auto_mode (scan + gated
launches); a Dataset of same-shaped variables (batched) == variable by
variable.  Round 6: entry-rich mappings run the shared 4 x 8 form; masked
`(Time, n, L)` cases also under REMAP_FLAG_BATCH_MASKS, with masks that do
and do not change in time; the auto mode is the layout-aware scan.

    python tools/fuzz_medium.py [seconds=480] [first_seed=0]
"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import assert_bitwise  # noqa: E402
from oracle import oracle  # noqa: E402
from pyremap_amd import engine, synthetic  # noqa: E402

LEVELS = [7, 16, 33, 48, 60, 61, 64, 65, 72, 80, 100, 101, 127, 130]


class Handle:
    """The opaque C plan handle (remap_plan_create / _apply / _destroy) on
    the same triplets: its own routing (C++) against the same oracle."""

    def __init__(self, m, dev, prepare):
        import ctypes
        self.ct = ctypes
        self.lib = engine.load_library()
        self.dev = dev
        self.n_a, self.n_b = m.n_a, m.n_b
        row = np.ascontiguousarray(m.row.cpu().numpy(), dtype=np.int32)
        col = np.ascontiguousarray(m.col.cpu().numpy(), dtype=np.int32)
        S = np.ascontiguousarray(m.S.cpu().numpy(), dtype=np.float64)
        fb = np.ascontiguousarray(m.frac_b.cpu().numpy(), dtype=np.float64)
        self.h = ctypes.c_void_p()
        dims = (ctypes.c_int64 * 2)(*m.dst_dims)
        rc = self.lib.remap_plan_create(
            m.n_b, m.n_a, len(S), row.ctypes.data, col.ctypes.data,
            S.ctypes.data, 1, fb.ctypes.data, 1, dims, 2, self._stream(),
            ctypes.byref(self.h))
        assert rc == 0, self.lib.remap_last_error()
        if prepare:
            assert self.lib.remap_plan_prepare_short_runs(
                self.h, self._stream()) == 0, self.lib.remap_last_error()

    def _stream(self):
        return self.ct.c_void_p(torch.cuda.current_stream().cuda_stream)

    def apply(self, x, shape, axis, masked, thr):
        lead = shape[:axis]
        tail = shape[axis + 1:]
        T = int(np.prod(lead)) if lead else 1
        L = int(np.prod(tail)) if tail else 1
        y = torch.empty(tuple(lead) + (self.n_b,) + tuple(tail),
                        dtype=torch.float64, device=self.dev)
        f = engine._Field()
        f.X, f.Y = x.data_ptr(), y.data_ptr()
        f.x_dtype = engine.DTYPE_F32 if x.dtype == torch.float32 \
            else engine.DTYPE_F64
        f.mode = engine.MODE_MASKED if masked else engine.MODE_FRACB
        f.threshold = float(thr or 0.0)
        f.n_batch, f.k_inner = T, L
        f.x_row_stride, f.x_batch_stride = L, self.n_a * L
        f.y_row_stride, f.y_batch_stride = L, self.n_b * L
        rc = self.lib.remap_plan_apply(self.h, self.ct.byref(f),
                                       self._stream())
        assert rc == 0, self.lib.remap_last_error()
        return y

    def close(self):
        self.lib.remap_plan_destroy(self.h)


def one(seed, dev):
    rng = np.random.default_rng(10_000 + seed)
    my, mx = int(rng.integers(100, 300)), int(rng.integers(100, 400))
    kind = rng.choice(['conservative', 'rich', 'bilinear'])
    if kind == 'bilinear':
        m = synthetic.bilinear_map((int(rng.integers(20, 120)),
                                    int(rng.integers(20, 150))), (my, mx),
                                   seed=seed, device=dev)
    else:
        lo, hi = (1, 7) if kind == 'conservative' else (10, 24)
        m = synthetic.conservative_map(
            int(rng.integers(5000, 80000)), (my, mx), lo, hi, seed=seed,
            device=dev, signed=kind == 'rich',
            locality=str(rng.choice(['raster', 'mesh', 'scatter'])))
    if rng.random() < 0.35:
        # a few LONG rows (pole caps of an ESMF-made map, or no structure at
        # all): applied apart, RemapPlan._split_long_rows
        import torch as _t
        n_long = int(rng.integers(1, 60))
        rows = rng.choice(m.n_b, n_long, replace=False)
        if rng.random() < 0.5:           # consecutive rows sharing a ring
            rows = (int(rows[0]) + np.arange(n_long)) % m.n_b
            ring = np.sort(rng.choice(m.n_a, int(rng.integers(100, min(
                1500, m.n_a))), replace=False))
            cols = [ring for _ in rows]
        else:
            cols = [rng.choice(m.n_a, int(rng.integers(97, min(900, m.n_a))),
                               replace=False) for _ in rows]
        add_row = np.concatenate([np.full(len(c), r + 1)
                                  for r, c in zip(rows, cols)])
        add_col = np.concatenate(cols) + 1
        add_S = rng.standard_normal(len(add_col)) / 50.0
        m.row = _t.cat([m.row, _t.as_tensor(add_row, dtype=m.row.dtype,
                                            device=dev)])
        m.col = _t.cat([m.col, _t.as_tensor(add_col, dtype=m.col.dtype,
                                            device=dev)])
        m.S = _t.cat([m.S, _t.as_tensor(add_S, device=dev)])
        m.frac_b[_t.as_tensor(rows, device=dev)] = 0.75
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, device=dev)
    sched = plan.auto_schedule(m.dst_dims) if rng.random() < 0.85 else None
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    src2d = None
    if kind == 'bilinear':
        src2d = m.src_dims
    handle = Handle(m, dev, prepare=bool(seed % 2)) if seed % 4 < 2 \
        else None
    multi = None
    if seed % 3 == 0:
        from pyremap_amd.parallel import MultiDeviceRemap
        multi = MultiDeviceRemap(plan, [dev] * int(rng.integers(2, 4)),
                                 grid_dims=m.dst_dims)
    for case in range(4):
        pick = rng.random()
        if src2d is not None and case == 3:
            # two source axes with other dims BETWEEN them, in place
            ny, nx = src2d
            lead = [int(rng.integers(1, 3))] if rng.random() < 0.5 else []
            between = [int(rng.integers(1, 9))]
            tail = [int(rng.integers(1, 5))] if rng.random() < 0.5 else []
            shape = lead + [ny] + between + [nx] + tail
            axes2 = [len(lead), len(lead) + 2]
            dtype = rng.choice([np.float64, np.float32])
            field = rng.standard_normal(shape).astype(dtype)
            masked = bool(rng.random() < 0.5)
            thr = None
            arg = field
            if masked:
                field[rng.random(shape) < 0.1] = np.nan
                thr = float(rng.choice([0.0, 0.05, 0.5]))
                arg = np.ma.masked_array(field, mask=np.isnan(field))
            ref = oracle.remap_numpy_array(csr, frac_b, m.dst_dims, arg,
                                           axes2, thr)
            ref = np.ma.filled(ref.astype(np.float64), np.nan) \
                if np.ma.isMaskedArray(ref) else np.asarray(ref)
            what = (f'seed {seed} case {case}: non-adjacent {shape} '
                    f'{np.dtype(dtype).name} masked={masked}')
            y = engine.remap_tensor(
                plan, m.dst_dims, torch.from_numpy(field).to(dev), axes2,
                engine.MODE_MASKED if masked else engine.MODE_FRACB,
                threshold=thr or 0.0)
            assert_bitwise(y.cpu().numpy(), ref, what)
            # ... and from / to host arrays
            from pyremap_amd import host_path
            got = host_path.remap_host_array(
                plan, m.dst_dims, field, axes2,
                mode='masked' if masked else 'fracb', threshold=thr).result()
            assert_bitwise(got, ref, what + ' host')
            continue
        if pick < 0.3:
            shape, axis = [m.n_a, int(rng.choice([40, 61, 128, 200, 257]))], 0
        elif pick < 0.5:
            shape, axis = [int(rng.integers(2, 200)), m.n_a], 1
        elif pick < 0.6:
            shape = [int(rng.integers(2, 40)), m.n_a, int(rng.integers(1, 8))]
            axis = 1
        else:
            shape = [int(rng.integers(1, 10)), m.n_a, int(rng.choice(LEVELS))]
            axis = 1
        dtype = rng.choice([np.float64, np.float32])
        field = rng.standard_normal(shape).astype(dtype)
        masked = bool(rng.random() < 0.5)
        arg, thr = field, None
        if masked:
            cells = rng.random(m.n_a) < 0.2
            # round 5: whole cells missing as often as values missing from
            # some column on (the per-row and the per-lane form of the
            # masked mode on entry-rich maps, spmm_groupmask.h), and both
            whole = rng.random()
            if axis == 0:
                if whole < 0.6:
                    field[rng.random(m.n_a) < 0.15] = np.nan
                if whole > 0.4:
                    field[cells, shape[1] // 2:] = np.nan
            elif len(shape) == 2:
                field[shape[0] // 2:, cells] = np.nan
            else:
                if whole < 0.5:
                    field[:, rng.random(m.n_a) < 0.15] = np.nan
                if whole > 0.3:
                    field[:, cells, shape[2] // 2:] = np.nan
                if shape[0] > 1 and rng.random() < 0.3:
                    # round 6: a mask that DOES change in time in a few
                    # cells (the time forms redo those groups)
                    field[int(rng.integers(0, shape[0])),
                          rng.random(m.n_a) < 0.02] = np.nan
            thr = float(rng.choice([0.0, 0.05, 0.5]))
            arg = np.ma.masked_array(field, mask=np.isnan(field))
        ref = oracle.remap_numpy_array(csr, frac_b, m.dst_dims, arg, [axis],
                                       thr)
        ref = np.ma.filled(ref.astype(np.float64), np.nan) \
            if np.ma.isMaskedArray(ref) else np.asarray(ref)
        x = torch.from_numpy(field).to(dev)
        emode = engine.MODE_MASKED if masked else engine.MODE_FRACB
        y = engine.remap_tensor(plan, m.dst_dims, x, [axis], emode,
                                threshold=thr or 0.0)
        what = (f'seed {seed} case {case}: {kind} {m.n_a}->{my}x{mx} '
                f'{sched and sched.get("family")} shape {shape} '
                f'{np.dtype(dtype).name} masked={masked} thr {thr}')
        assert tuple(y.shape) == ref.shape, what
        assert_bitwise(y.cpu().numpy(), ref, what)
        if masked:
            # the hint changes the kernel form, never the bits; and the
            # device-side choice (scan + gated launches) is the reference's
            yf = engine.remap_tensor(plan, m.dst_dims, x, [axis], emode,
                                     threshold=thr or 0.0,
                                     flags=engine.FLAG_CELL_MASKS)
            assert_bitwise(yf.cpu().numpy(), ref, what + ' cell-masks hint')
            if len(shape) == 3:
                # round 6: "the mask does not change from batch to batch"
                # (spmm_timeshare / spmm_grouptime; a hint as well)
                yb = engine.remap_tensor(plan, m.dst_dims, x, [axis], emode,
                                         threshold=thr or 0.0,
                                         flags=engine.FLAG_BATCH_MASKS)
                assert_bitwise(yb.cpu().numpy(), ref,
                               what + ' batch-masks hint')
            if np.isnan(field).any():
                ya = engine.remap_tensor_auto_mode(plan, m.dst_dims, x,
                                                   [axis], thr)
                assert_bitwise(ya.cpu().numpy(), ref, what + ' auto mode')
        if handle is not None:
            yh = handle.apply(x, shape, axis, masked, thr)
            assert_bitwise(yh.cpu().numpy().reshape(ref.shape), ref,
                           what + ' plan handle')
        if case == 1:
            # the same through the host-array path (numpy in, numpy out)
            from pyremap_amd import host_path
            got = host_path.remap_host_array(
                plan, m.dst_dims, field, [axis],
                mode='masked' if masked else 'fracb', threshold=thr).result()
            assert_bitwise(got, ref, what + ' host')
        r = int(rng.integers(0, 3))
        shard = plan.shard(r, 3)
        shard.auto_schedule(m.dst_dims)
        ys = engine.remap_tensor(shard, None, x, [axis], emode,
                                 threshold=thr or 0.0)
        lead = shape[:axis]
        flat = ref.reshape(tuple(lead) + (m.n_b,) + tuple(shape[axis + 1:]))
        lo = shard.row_offset
        want = np.take(flat, np.arange(lo, lo + shard.n_b), axis=axis)
        assert_bitwise(ys.cpu().numpy(), want, what + f' shard {r}/3')
        packed, ucols = shard.packed()
        packed.auto_schedule(m.dst_dims)
        yp = engine.remap_tensor(packed, None,
                                 engine.gather_rows(x, axis, ucols), [axis],
                                 emode, threshold=thr or 0.0)
        assert_bitwise(yp.cpu().numpy(), want, what + f' packed {r}/3')
        if multi is not None:
            ym = engine.remap_tensor(multi, m.dst_dims, x, [axis], emode,
                                     threshold=thr or 0.0)
            assert_bitwise(ym.cpu().numpy(), ref, what + ' multi-device')
    if handle is not None:
        handle.close()


def dataarray_level(seed, dev):
    """`Remapper.remap_numpy(DataArray)` on random dims orders and dtypes --
    the xarray-level bookkeeping of remap_numpy.py:150-220 (which dims the
    result has, where the destination dims go, float64 out, masked iff a NaN
    and a threshold) -- against the oracle's array-level result."""
    from pyremap_amd import DataArray, Remapper
    rng = np.random.default_rng(77_000 + seed)
    two_d = bool(rng.random() < 0.5)
    if two_d:
        m = synthetic.bilinear_map((int(rng.integers(8, 30)),
                                    int(rng.integers(8, 30))),
                                   (int(rng.integers(10, 60)),
                                    int(rng.integers(10, 60))), seed=seed,
                                   device=dev)
        src_dims = ['y', 'x']
    else:
        m = synthetic.conservative_map(
            int(rng.integers(500, 5000)),
            (int(rng.integers(10, 60)), int(rng.integers(10, 60))), 1, 6,
            seed=seed, device=dev,
            locality=str(rng.choice(['raster', 'mesh'])))
        src_dims = ['nCells']

    class Desc:
        pass
    src, dst = Desc(), Desc()
    src.dims, src.dim_sizes = src_dims, list(m.src_dims)
    dst.dims, dst.dim_sizes = ['lat', 'lon'], list(m.dst_dims)
    dst.coords, dst.mesh_name = {}, 'fuzz'
    mm = m.numpy()
    r = Remapper.from_triplets(mm['row'], mm['col'], mm['S'], mm['frac_b'],
                               src, dst, device=dev)
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    for case in range(4):
        n_extra = int(rng.integers(0, 4))
        extras = [(f'e{i}', int(rng.integers(1, 7))) for i in range(n_extra)]
        # the source dims keep their relative order; extras go anywhere
        dims = list(zip(src_dims, m.src_dims))
        for e in extras:
            dims.insert(int(rng.integers(0, len(dims) + 1)), e)
        names = [d[0] for d in dims]
        shape = [int(d[1]) for d in dims]
        kind = rng.choice(['f8', 'f4', 'i4'])
        if kind == 'i4':
            data = rng.integers(-50, 50, size=shape).astype(np.int32)
        else:
            data = rng.standard_normal(shape).astype(kind)
            if rng.random() < 0.5:
                data[rng.random(shape) < 0.1] = np.nan
        thr = None if rng.random() < 0.4 else \
            float(rng.choice([0.0, 0.1, 0.6]))
        out = r.remap_numpy(DataArray(data, dims=names, name='v'), thr)
        axes = [names.index(d) for d in src_dims]
        has_nan = data.dtype.kind == 'f' and bool(np.isnan(data).any())
        arg = np.ma.masked_array(data, np.isnan(data)) if has_nan else data
        ref = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims, arg,
                                       axes, thr)
        ref = np.ma.filled(np.ma.masked_array(ref).astype(np.float64),
                           np.nan)
        first = axes[0]
        want_dims = names[:first] + ['lat', 'lon'] + \
            [d for d in names[first:] if d not in src_dims]
        what = (f'seed {seed} DataArray {list(zip(names, shape))} {kind} '
                f'thr {thr} nan {has_nan}')
        assert list(out.dims) == want_dims, what
        assert out.values.dtype == np.float64, what
        assert_bitwise(out.values, ref, what)
    # round 5: a Dataset of several same-shaped variables (travel and are
    # remapped together, host_path.remap_host_batch) == each on its own
    from pyremap_amd import Dataset
    from pyremap_amd.remapper import remap_numpy as rn
    lead = [('Time', int(rng.integers(1, 4)))] if rng.random() < 0.7 else []
    tail = [('nLev', int(rng.integers(1, 6)))] if rng.random() < 0.4 else []
    dims = lead + list(zip(src_dims, m.src_dims)) + tail
    names = [d[0] for d in dims]
    shape = [int(d[1]) for d in dims]
    ds = Dataset()
    for v in range(int(rng.integers(2, 12))):
        data = rng.standard_normal(shape).astype(
            rng.choice(['f8', 'f8', 'f4']))
        if rng.random() < 0.5:
            data[rng.random(shape) < 0.1] = np.nan
        ds[f'v{v}'] = DataArray(data, dims=names)
    thr = None if rng.random() < 0.4 else float(rng.choice([0.0, 0.1, 0.6]))
    out = r.remap_numpy(ds, thr)
    for name in ds.data_vars:
        alone = rn._remap_data_array(ds[name], r, thr)
        assert list(out[name].dims) == list(alone.dims)
        assert_bitwise(out[name].values, alone.values,
                       f'seed {seed} Dataset {name} {shape} thr {thr}')


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 480.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device('cuda', 0)
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < budget and len(bad) < 4:
        try:
            one(seed, dev)
            dataarray_level(seed, dev)
            n += 1
        except Exception as exc:   # noqa: BLE001 - reported
            print('SEED', seed, 'FAILED', type(exc).__name__, str(exc)[:400])
            bad.append(seed)
        seed += 1
    print(f'seeds ok: {n}, failed: {bad}, next seed {seed}, '
          f'{time.time() - t0:.0f} s')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
