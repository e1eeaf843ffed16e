"""
Rows sharded over several GPUs, on the one
GPU a test box has:

* a shard in its PACKED column space (``RemapPlan.packed`` ->
  ``remap_pack_columns``) applied to ``X[ucols]`` (``remap_gather_rows``) is
  the unsharded result bit for bit -- on a source mesh numbered as MPAS
  numbers its cells, where a (min, max) band of source rows is all of X;
* ``MultiDeviceRemap`` / ``Remapper(devices=[...])`` -- ONE process driving N
  devices, here the same GPU listed N times -- returns the same arrays,
  Datasets and masks as the one-device Remapper;
* ``ShardedRemap`` / ``Remapper.use_process_group`` -- one process per rank
  (gloo rendezvous, the ranks sharing this GPU) -- every exchange form,
  pipelined batches, and ``remap_numpy`` as a collective;
* the same through RCCL when the box has two GPUs (skipped otherwise).
"""
import os
import socket

import numpy as np
import pytest

from helpers import assert_bitwise

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from pyremap_amd import engine
    engine.load_library()
    return torch.device('cuda', 0)


def _mesh_problem(dev, n_a=30000, dims=(120, 200), seed=21):
    from pyremap_amd import engine, synthetic
    m = synthetic.conservative_map(n_a, dims, 2, 7, seed=seed, device=dev,
                                   locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    return m, plan


def test_pack_columns_and_gather_rows(dev):
    from pyremap_amd import engine
    m, plan = _mesh_problem(dev)
    sl = plan.row_slice(5000, 11000)
    packed, ucols = sl.packed()
    want_u = torch.unique(sl.col.to(torch.int64))
    assert ucols.dtype == torch.int32
    assert torch.equal(ucols.to(torch.int64), want_u)
    assert packed.n_a == want_u.numel() < plan.n_a
    assert torch.equal(ucols[packed.col.to(torch.int64)], sl.col)
    # one destination-row shard meets source ids from most of the range: the
    # (min, max) band round 2 sent is all of X, the packed rows are not
    assert int(ucols.max() - ucols.min()) > 0.9 * plan.n_a
    assert ucols.numel() < 0.5 * plan.n_a
    # gather_rows == index_select, every unit width and layout
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    for shape, axis, dtype in (((plan.n_a, 64), 0, torch.float64),
                               ((3, plan.n_a, 61), 1, torch.float64),
                               ((12, plan.n_a), 1, torch.float64),
                               ((plan.n_a,), 0, torch.float32),
                               ((2, plan.n_a, 5), 1, torch.float32),
                               ((plan.n_a, 130), 0, torch.float32),
                               ((2, 2, plan.n_a, 7), 2, torch.float64)):
        x = torch.randn(shape, generator=g, device=dev, dtype=torch.float64)
        x = x.to(dtype)
        got = engine.gather_rows(x, axis, ucols)
        assert torch.equal(got, x.index_select(axis, ucols.to(torch.int64)))
    odd = torch.randn(plan.n_a * 3 + 1, generator=g, device=dev,
                      dtype=torch.float32)[1:].reshape(plan.n_a, 3)
    assert torch.equal(engine.gather_rows(odd, 0, ucols),
                       odd.index_select(0, ucols.to(torch.int64)))
    assert engine.gather_rows(x, 2, ucols[:0]).shape == (2, 2, 0, 7)


@pytest.mark.parametrize('mode', ['fracb', 'masked'])
def test_packed_shards_are_the_unsharded_result(dev, mode):
    from pyremap_amd import engine
    m, plan = _mesh_problem(dev)
    plan.auto_schedule(m.dst_dims)
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    emode = engine.MODE_MASKED if mode == 'masked' else engine.MODE_FRACB
    for shape, axes, row_axis in (((plan.n_a, 200), [0], 0),
                                  ((3, plan.n_a, 64), [1], 1),
                                  ((2, plan.n_a, 61), [1], 1),
                                  ((20, plan.n_a), [1], 1)):
        x = torch.randn(shape, generator=g, device=dev, dtype=torch.float64)
        if mode == 'masked':
            dead = torch.rand(plan.n_a, generator=g, device=dev) < 0.2
            x.index_fill_(axes[0], dead.nonzero().squeeze(1), float('nan'))
        want = engine.remap_tensor(plan, None, x, axes, emode, threshold=0.01)
        slabs = []
        for r in range(5):
            sl = plan.shard(r, 5)
            packed, ucols = sl.packed()
            packed.auto_schedule(m.dst_dims)
            slabs.append(engine.remap_tensor(
                packed, None, engine.gather_rows(x, axes[0], ucols), axes,
                emode, threshold=0.01))
        assert_bitwise(torch.cat(slabs, dim=row_axis).cpu().numpy(),
                       want.cpu().numpy(), f'{mode} {shape}')


def test_scan_nan_takes_unaligned_views(dev):
    """a contiguous view that starts inside an allocation
    (big[1:]) is element-aligned only."""
    from pyremap_amd import engine
    for dtype in (torch.float64, torch.float32):
        for start in (1, 2, 3):
            for n in (1, 2, 5, 1000, 100003):
                big = torch.zeros(n + start, device=dev, dtype=dtype)
                for pos in (None, 0, n - 1, n // 2):
                    big.zero_()
                    if pos is not None:
                        big[start + pos] = float('nan')
                    big[0] = float('nan')        # in FRONT of the view
                    flag = torch.zeros(1, dtype=torch.int32, device=dev)
                    engine.scan_nan(big[start:], flag)
                    assert int(flag) == int(pos is not None), \
                        (dtype, start, n, pos)


class _Desc:
    def __init__(self, dims, sizes, name='d'):
        self.dims = list(dims)
        self.dim_sizes = [int(s) for s in sizes]
        self.coords = {}
        self.mesh_name = name


def _remappers(m, dev, n_dev):
    from pyremap_amd import Remapper
    mm = m.numpy()
    src, dst = _Desc(['nCells'], [m.n_a]), _Desc(['lat', 'lon'], m.dst_dims)
    one = Remapper.from_triplets(mm['row'], mm['col'], mm['S'], mm['frac_b'],
                                 src, dst, device=dev)
    many = Remapper.from_triplets(mm['row'], mm['col'], mm['S'],
                                  mm['frac_b'], src, dst, device=dev)
    many.devices = [dev] * n_dev
    return one, many


@pytest.mark.parametrize('n_dev', [2, 3])
def test_remapper_devices_equals_one_device(dev, n_dev):
    """`Remapper(devices=[...])` (remapper.py:508-532's call shape, one
    process): identical arrays, masks and Datasets from 1 and N devices."""
    from pyremap_amd import Dataset, DataArray, parallel
    m, _ = _mesh_problem(dev, n_a=12000, dims=(60, 90), seed=5)
    one, many = _remappers(m, dev, n_dev)
    rng = np.random.default_rng(n_dev)
    # arrays: numpy in -> masked array out
    for shape, axes in (((m.n_a, 40), [0]), ((3, m.n_a, 17), [1]),
                        ((m.n_a,), [0]), ((6, m.n_a), [1]),
                        ((2, m.n_a, 3, 4), [1])):
        x = rng.standard_normal(shape)
        for thr, nan in ((None, False), (0.05, True), (0.05, False)):
            xx = x.copy()
            if nan:
                dead = rng.random(m.n_a) < 0.15
                xx[(slice(None),) * axes[0] + (dead,)] = np.nan
            arg = np.ma.masked_array(xx, np.isnan(xx)) if nan else xx
            a = one.remap_array(arg, axes, thr)
            b = many.remap_array(arg, axes, thr)
            assert isinstance(many._matrix, parallel.MultiDeviceRemap)
            assert len(many._matrix.shards) == n_dev
            assert np.array_equal(np.ma.getmaskarray(a),
                                  np.ma.getmaskarray(b))
            assert_bitwise(np.ma.filled(b, np.nan), np.ma.filled(a, np.nan),
                           f'{shape} thr={thr} nan={nan}')
    # device tensors: nothing leaves the GPU(s)
    xt = torch.from_numpy(rng.standard_normal((4, m.n_a, 33))).to(dev)
    xt[:, torch.rand(m.n_a, device=dev) < 0.1, :] = float('nan')
    for thr in (None, 0.02):
        a = one.remap_array(xt, [1], thr)
        b = many.remap_array(xt, [1], thr)
        assert b.device == a.device
        assert_bitwise(b.cpu().numpy(), a.cpu().numpy(), f'tensor thr={thr}')
    # non-adjacent source axes do not exist for a 1-D mesh; a 2-D source does
    # a Dataset, with a threshold, NaNs in some variables only
    data = {
        'ssh': DataArray(rng.standard_normal((2, m.n_a)),
                         dims=('Time', 'nCells'), attrs={'units': 'm'}),
        'temp': DataArray(np.where(rng.random((2, m.n_a, 5)) < 0.1, np.nan,
                                   rng.standard_normal((2, m.n_a, 5))),
                          dims=('Time', 'nCells', 'nVertLevels')),
        'f32': DataArray(rng.standard_normal((m.n_a, 9)).astype(np.float32),
                         dims=('nCells', 'k')),
        'untouched': DataArray(np.arange(5.0), dims=('nVertLevels',)),
    }
    ds = Dataset(data, attrs={'title': 't'})
    for thr in (None, 0.01):
        a = one.remap_numpy(ds, thr)
        b = many.remap_numpy(ds, thr)
        assert list(a.data_vars) == list(b.data_vars)
        for name in a.data_vars:
            assert a[name].dims == b[name].dims
            assert_bitwise(b[name].values, a[name].values, f'{name} {thr}')


def test_multi_device_2d_source_and_slabs(dev):
    """A 2-D (lat, lon) source, adjacent and NON-adjacent source axes, and
    the slabs left on their devices (`gather=False`)."""
    from pyremap_amd import engine, synthetic
    from pyremap_amd.parallel import MultiDeviceRemap
    m = synthetic.bilinear_map((30, 40), (70, 90), device=dev)
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    multi = MultiDeviceRemap(plan, [dev, dev, dev], grid_dims=m.dst_dims)
    plan.auto_schedule(m.dst_dims)
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    for shape, axes in (((30, 40, 50), [0, 1]), ((4, 30, 40, 9), [1, 2]),
                        ((30, 7, 40), [0, 2]), ((3, 30, 5, 40, 2), [1, 3])):
        x = torch.randn(shape, generator=g, device=dev, dtype=torch.float64)
        want = engine.remap_tensor(plan, m.dst_dims, x, axes,
                                   engine.MODE_FRACB)
        got, mask = engine.remap_tensor(multi, m.dst_dims, x, axes,
                                        engine.MODE_FRACB, want_mask=True)
        assert_bitwise(got.cpu().numpy(), want.cpu().numpy(), str(shape))
        assert not bool(mask.any())
    x = torch.randn((2, 30, 40, 16), generator=g, device=dev,
                    dtype=torch.float64)
    slabs = multi.remap_tensor(m.dst_dims, x, [1, 2], engine.MODE_FRACB,
                               gather=False)
    want = engine.remap_tensor(plan, None, x.reshape(2, 1200, 16), [1],
                               engine.MODE_FRACB)
    assert [s.shape[1] for s in slabs] == \
        [b - a for a, b in zip(multi.bounds, multi.bounds[1:])]
    assert_bitwise(torch.cat(slabs, 1).cpu().numpy(), want.cpu().numpy(),
                   'slabs')
    assert multi.packed_fraction() < 0.7
    with pytest.raises(ValueError):
        engine.remap_tensor(multi, m.dst_dims, x, [1, 2], engine.MODE_FRACB,
                            out=want)


# ---------------------------------------------------------------------------
# one process per rank
# ---------------------------------------------------------------------------

def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _sharded_worker(rank, world, port, tmpdir, backend):
    import sys

    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    local = rank if backend == 'nccl' else 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle
        from pyremap_amd import Dataset, DataArray, Remapper, engine, \
            synthetic
        from pyremap_amd.parallel import ShardedRemap
        m = synthetic.conservative_map(4000, (40, 60), 1, 7, seed=44,
                                       locality='mesh')
        mm = m.numpy()
        full = engine.RemapPlan.from_triplets(
            mm['row'], mm['col'], mm['S'], mm['frac_b'], m.n_a, m.n_b,
            device=dev)
        csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'],
                                m.n_b, m.n_a)
        K = 96
        rng = np.random.default_rng(3)
        host = [rng.standard_normal((m.n_a, K)) for _ in range(3)]
        notes = {}
        refs = []
        for h in host:
            ref, ref_mask = oracle.remap_flat(csr, mm['frac_b'], h, False,
                                              0.0)
            ref[ref_mask] = np.nan
            refs.append(ref)
        # gloo moves GPU tensors through the host; RCCL does not
        hows = ('alltoall', 'broadcast')
        for how in hows:
            sharded = ShardedRemap(full, grid_dims=m.dst_dims, exchange=how)
            assert sharded.exchange == how
            notes['packed_frac'] = sharded.packed_fraction()
            # one field: only rank 0 holds it
            x = torch.from_numpy(host[0]).to(dev) if rank == 0 else None
            packed = sharded.distribute(x, src=0, shape=(m.n_a, K),
                                        dtype=torch.float64)
            assert packed.shape == (sharded.ucols.shape[0], K)
            y = sharded.gather(sharded.apply(packed, [0],
                                             engine.MODE_FRACB))
            notes[how] = int(np.array_equal(y.cpu().numpy(), refs[0],
                                            equal_nan=True))
            # three batches pipelined: b + 1 travels while b is computed
            batches = [torch.from_numpy(h).to(dev) if rank == 0 else
                       torch.full((m.n_a, K), float('nan'),
                                  dtype=torch.float64, device=dev)
                       for h in host]
            outs = sharded.apply_pipelined(batches, engine.MODE_FRACB,
                                           how=how)
            r0, r1 = sharded.bounds[rank], sharded.bounds[rank + 1]
            ok = 1
            for ref, o in zip(refs, outs):
                ok &= int(np.array_equal(o.cpu().numpy(), ref[r0:r1],
                                         equal_nan=True))
            notes['pipelined_' + how] = ok
            notes['schedule'] = sharded.schedule['family']
        # Remapper as a collective: every rank the same call, rank 0's data
        class Desc:
            pass
        src_d, dst_d = Desc(), Desc()
        src_d.dims, src_d.dim_sizes = ['nCells'], [m.n_a]
        dst_d.dims, dst_d.dim_sizes = ['lat', 'lon'], list(m.dst_dims)
        dst_d.coords, dst_d.mesh_name = {}, 'grid'
        r = Remapper.from_triplets(mm['row'], mm['col'], mm['S'],
                                   mm['frac_b'], src_d, dst_d, device=dev)
        r.use_process_group(src=0)
        field = rng.standard_normal((2, m.n_a, 5))
        field[:, rng.random(m.n_a) < 0.2, :] = np.nan
        mine = field if rank == 0 else np.zeros_like(field)   # shapes only
        ds = Dataset({'t': DataArray(mine, dims=('Time', 'nCells', 'z')),
                      's': DataArray(mine[0, :, 0], dims=('nCells',))})
        out = r.remap_numpy(ds, 0.01)
        arg = np.ma.masked_array(field, np.isnan(field))
        want = np.ma.filled(oracle.remap_numpy_array(
            csr, mm['frac_b'], m.dst_dims, arg, [1], 0.01), np.nan)
        notes['collective_dataset'] = int(
            np.array_equal(out['t'].values, want, equal_nan=True) and
            out['t'].dims == ('Time', 'lat', 'lon', 'z') and
            np.array_equal(out['s'].values, want[0, :, :, 0],
                           equal_nan=True))
        got = r.remap_array(mine, [1], None)
        want = np.ma.filled(oracle.remap_numpy_array(
            csr, mm['frac_b'], m.dst_dims, field, [1], None), np.nan)
        notes['collective_array'] = int(np.array_equal(
            np.ma.filled(got, np.nan), want, equal_nan=True))
        # _remap_numpy_array under the group: the reference's values AND
        # mask (:262-278) -- a MaskedArray whose mask hides finite values,
        # with one NaN that is NOT masked (it goes through, :263: every cell
        # it touches is NaN and unmasked); the frac_b branch with NaNs (mask
        # = frac_b <= 0 only, NaNs propagate unmasked); a plain ndarray
        from pyremap_amd.remapper.remap_numpy import _remap_numpy_array
        data = rng.standard_normal((2, m.n_a, 3))
        hide = rng.random(data.shape) < 0.3
        data[0, 17, 1] = np.nan
        hide[0, 17, 1] = False
        data[1, 40:60, :] = np.nan
        hide[1, 40:60, :] = True
        ok = 1
        for arg, thr in ((np.ma.masked_array(data, hide), 0.01),
                         (np.ma.masked_array(data, hide), None),
                         (data, 0.01), (np.nan_to_num(data), None)):
            if rank == 0:
                mine = arg
            elif isinstance(arg, np.ma.MaskedArray):   # shapes and type only
                mine = np.ma.masked_array(np.zeros_like(data),
                                          np.zeros(data.shape, bool))
            else:
                mine = np.zeros_like(data)
            got = _remap_numpy_array(r, mine, [1], thr)
            want = oracle.remap_numpy_array(csr, mm['frac_b'], m.dst_dims,
                                            arg, [1], thr)
            wm, gm = np.ma.getmaskarray(want), np.ma.getmaskarray(got)
            ok &= int(isinstance(got, np.ma.MaskedArray))
            ok &= int(np.array_equal(gm, wm))
            ok &= int(np.array_equal(np.ma.getdata(got)[~wm],
                                     np.ma.getdata(want)[~wm],
                                     equal_nan=True))
        notes['collective_masks'] = ok
        # ncremap as a collective: every rank reads, rank 0 alone writes
        from pyremap_amd.io.netcdf import open_dataset, write_netcdf
        src_path = os.path.join(tmpdir, 'collective_in.nc')
        out_path = os.path.join(tmpdir, 'collective_out.nc')
        if rank == 0:
            write_netcdf(Dataset({'t': DataArray(
                field, dims=('Time', 'nCells', 'z'))}), src_path,
                format='NETCDF3_64BIT_DATA')
        dist.barrier()
        r.ncremap(src_path, out_path, renormalize=0.01)
        dist.barrier()
        if rank == 0:
            back = open_dataset(out_path)
            arg = np.ma.masked_array(field, np.isnan(field))
            want = np.ma.filled(oracle.remap_numpy_array(
                csr, mm['frac_b'], m.dst_dims, arg, [1], 0.01), np.nan)
            notes['collective_file'] = int(np.array_equal(
                back['t'].values, want, equal_nan=True))
        else:
            notes['collective_file'] = 1
        with open(os.path.join(tmpdir, f'rank{rank}.txt'), 'w') as f:
            f.write(repr(notes))
    finally:
        dist.destroy_process_group()


def _run_ranks(world, tmp_path, backend):
    import torch.multiprocessing as mp
    mp.spawn(_sharded_worker,
             args=(world, _free_port(), str(tmp_path), backend),
             nprocs=world, join=True)
    for rank in range(world):
        notes = eval(open(tmp_path / f'rank{rank}.txt').read())
        for key in ('alltoall', 'broadcast', 'pipelined_alltoall',
                    'pipelined_broadcast', 'collective_dataset',
                    'collective_array', 'collective_masks',
                    'collective_file'):
            assert notes[key] == 1, (rank, key, notes)
        # mesh-numbered source: the packed rows still are ~(1/N + halo)
        assert notes['packed_frac'] < {1: 1.01, 2: 0.8}.get(world, 0.65)


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_remap_ranks_sharing_one_gpu(world, tmp_path):
    """
    `ShardedRemap` and `Remapper.use_process_group` end to end on the HIP
    kernels with `world` processes (gloo rendezvous, all on this box's one
    GPU): per-rank packed plans and schedules, the field delivered as packed
    rows (all_to_all_single) and as one broadcast + local gather, three
    batches pipelined, `remap_numpy` as a collective -- every value the
    oracle's.
    """
    _run_ranks(world, tmp_path, 'gloo')


def test_sharded_remap_on_the_rccl_backend_one_rank(tmp_path):
    """Every collective the sharded path issues -- broadcast,
    all_to_all_single with uneven splits, all_gather of the slabs, the
    pipelined batches, `Remapper.use_process_group` -- through the REAL
    backend (`nccl` = RCCL, GPU tensors, no host staging) with the one rank
    a 1-GPU box can hold: argument checks, dtypes, stream ordering between
    RCCL's stream and the launch stream."""
    _run_ranks(1, tmp_path, 'nccl')


@pytest.mark.skipif(torch.cuda.device_count() < 2,
                    reason='needs two GPUs (RCCL over xGMI)')
def test_sharded_remap_over_rccl(tmp_path):
    """The same through RCCL, one rank per GPU -- the run the advisor asked
    for before the all-to-all exchange becomes the nccl default."""
    _run_ranks(2, tmp_path, 'nccl')


@pytest.mark.skipif(torch.cuda.device_count() < 2,
                    reason='needs two physical GPUs (peer copies)')
def test_two_physical_gpus_stress_under_a_side_stream():
    """
    `Remapper(devices=[cuda:0, cuda:1])` on two PHYSICAL GPUs (the other
    tests list one GPU N times): remap_tensor looped under a non-default
    stream with allocator churn between the calls -- a missing dependency
    between the gather on the source stream, the peer copy, the launch on the
    other device's stream and the copy back into the assembled result shows
    up as a stale or torn slab.  Every call bitwise the one-device result.
    """
    from pyremap_amd import engine, synthetic
    from pyremap_amd.parallel import MultiDeviceRemap
    d0, d1 = torch.device('cuda', 0), torch.device('cuda', 1)
    m = synthetic.conservative_map(30000, (120, 200), 2, 7, seed=5,
                                   device=d0, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                          m.n_b, index_base=1, device=d0)
    plan.auto_schedule(m.dst_dims)
    multi = MultiDeviceRemap(plan, [d0, d1], grid_dims=m.dst_dims)
    side = torch.cuda.Stream(device=d0)
    g = torch.Generator(device=d0)
    g.manual_seed(9)
    for it in range(40):
        K = (64, 96, 130)[it % 3]
        with torch.cuda.stream(side):
            x = torch.randn((m.n_a, K), generator=g, device=d0,
                            dtype=torch.float64)
            junk = [torch.empty(1 << (18 + it % 5), device=d)
                    for d in (d0, d1)]              # allocator churn
            y = multi.remap_tensor(m.dst_dims, x, [0], engine.MODE_FRACB)
            ref = engine.remap_tensor(plan, m.dst_dims, x, [0],
                                      engine.MODE_FRACB)
            del junk
        side.synchronize()
        assert torch.equal(torch.nan_to_num(y, nan=-7.0),
                           torch.nan_to_num(ref, nan=-7.0)), it


def test_shards_without_entries(dev):
    """A shard that holds nothing but empty destination rows (land): its
    packed column space is empty, its slab all masked -- and the assembled
    result still is the one-device result."""
    from pyremap_amd import engine, synthetic
    from pyremap_amd.parallel import MultiDeviceRemap
    m = synthetic.conservative_map(3000, (40, 50), 1, 5, seed=12, device=dev,
                                   locality='mesh')
    keep = m.row > 1200                      # rows 0..1199: no entries
    frac_b = m.frac_b.clone()
    frac_b[:1200] = 0.0
    plan = engine.RemapPlan.from_triplets(m.row[keep], m.col[keep],
                                          m.S[keep], frac_b, m.n_a, m.n_b,
                                          device=dev)
    empty, ucols = plan.row_slice(0, 1000).packed()
    assert empty.nnz == 0 and empty.n_a == 0 and ucols.numel() == 0
    multi = MultiDeviceRemap(plan, [dev] * 4, grid_dims=m.dst_dims)
    assert any(s.plan.nnz == 0 for s in multi.shards) or \
        multi.bounds[1] <= 1200
    plan.auto_schedule(m.dst_dims)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for shape, axes in (((m.n_a, 70), [0]), ((3, m.n_a, 9), [1]),
                        ((50, m.n_a), [1])):
        x = torch.randn(shape, generator=g, device=dev, dtype=torch.float64)
        x.index_fill_(axes[0], torch.arange(0, m.n_a, 7, device=dev),
                      float('nan'))
        for mode, thr in ((engine.MODE_FRACB, 0.0),
                          (engine.MODE_MASKED, 0.05)):
            want = engine.remap_tensor(plan, m.dst_dims, x, axes, mode,
                                       threshold=thr)
            got = engine.remap_tensor(multi, m.dst_dims, x, axes, mode,
                                      threshold=thr)
            assert_bitwise(got.cpu().numpy(), want.cpu().numpy(),
                           f'{shape} {mode}')


def test_entry_rich_shards_take_the_shared_and_the_masked_forms(dev):
    """Round 6: a packed row shard of an entry-rich mapping schedules itself
    like the whole mapping -- the shared lists (`spmm_groupshare`), and under
    `remap_tensor_auto_mode` the masked form the layout-aware scan names
    (whole cells / the same mask in every batch / neither) -- and N shards
    give the bits of the one-device plan, which are the oracle's
    (remap_numpy.py:201-204, 258-278)."""
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    from pyremap_amd.parallel import MultiDeviceRemap
    m = synthetic.conservative_map(4000, (60, 80), 10, 24, seed=4,
                                   device=dev, signed=True, locality='mesh')
    plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                          m.n_a, m.n_b, device=dev)
    plan.auto_schedule(m.dst_dims)
    multi = MultiDeviceRemap(plan, [dev, dev, dev], grid_dims=m.dst_dims)
    for shard in multi.shards:
        g = shard.plan.groups
        assert g is not None and g['rows'] == 8 and 'share' in g
        assert shard.plan.default_tune[engine.MODE_FRACB][5] == 32
    rowptr, col, val = plan.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
    frac_b = m.frac_b.cpu().numpy()
    rng = np.random.default_rng(6)
    T, L = 6, 64
    base = rng.standard_normal((T, m.n_a, L))
    land = base.copy()
    land[:, rng.random(m.n_a) < 0.25, :] = np.nan
    depth = rng.integers(1, L + 1, m.n_a)
    bathy = land.copy()
    bathy[:, np.arange(L)[None, :] >= depth[:, None]] = np.nan
    vary = bathy.copy()
    vary[3, 17, 2] = np.nan
    for tag, f in (('no NaN', base), ('land', land), ('bathymetry', bathy),
                   ('varying', vary)):
        fd = torch.from_numpy(f).to(dev)
        one = engine.remap_tensor_auto_mode(plan, m.dst_dims, fd, [1], 0.3)
        many = multi.remap_tensor_auto_mode(m.dst_dims, fd, [1], 0.3)
        flat = np.ascontiguousarray(f.transpose(1, 0, 2)).reshape(m.n_a, -1)
        masked = bool(np.isnan(flat).any())
        ref, ref_mask = oracle.remap_flat(csr, frac_b, flat, masked, 0.3)
        ref = ref.copy()
        ref[ref_mask] = np.nan
        ref = ref.reshape(m.n_b, T, L).transpose(1, 0, 2)
        assert_bitwise(one.cpu().numpy().reshape(ref.shape), ref,
                       f'one device, {tag}')
        assert_bitwise(many.cpu().numpy().reshape(ref.shape), ref,
                       f'three shards, {tag}')
    # the frac_b mode on (n_a, K): the shared form per shard
    x = rng.standard_normal((m.n_a, 384))
    xd = torch.from_numpy(x).to(dev)
    ref, ref_mask = oracle.remap_flat(csr, frac_b, x, False, 0.0)
    ref = ref.copy()
    ref[ref_mask] = np.nan
    y = multi.remap_tensor(m.dst_dims, xd, [0], engine.MODE_FRACB)
    assert_bitwise(y.cpu().numpy().reshape(m.n_b, -1), ref, 'shards, frac_b')
