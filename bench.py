#!/usr/bin/env python3
"""
bench.py -- the reference's headline workload on MI355X.

Metric (BASELINE.json): dst cells/sec + HBM GB/s, EC30to60 MPAS -> 0.5 deg
lat-lon, 512 batched fp64 fields.  A "step" is ONE pass of the hot path
(`Remapper.remap_numpy`'s weight application: CSR SpMM + frac_b
normalisation + masking, one fused HIP launch through the C ABI) over one
batch of 512 synthetic fields that are already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU): destination
rows are sharded over the ranks (nnz-balanced contiguous ranges), every
shard lives in the compact space of the source rows it references
(`RemapPlan.packed`), rank 0's source field reaches the ranks ONCE before the
timed region (one RCCL broadcast, timed separately: `multi_gpu.broadcast_ms`,
then each rank gathers its packed rows), and the timed steps contain no
collective.  After the metric, under a watchdog, `multi_gpu.packed_ms` times
the exchange in which each rank receives ONLY its packed rows (one
all_to_all_single) and `multi_gpu.pipelined_*` the whole job WITH the
exchange, K-chunks pipelined behind the kernel.  The problem size is fixed,
so this is strong scaling.

The source cells of the synthetic conservative maps are numbered the way an
MPAS mesh numbers its cells (`synthetic.mesh_numbering`, calibrated to the
reference's QU240 fixture), not along the destination raster;
`roofline.workloads.config3_raster_numbering` is the same map in the raster
numbering round 2 measured.

Prints ONE JSON line on rank 0.  `value` = destination cell-fields per second
for the whole job; `roofline` prices the kernel against HBM bandwidth using
SURVEY.md section 8(d)'s algorithmic bytes; `cpu_baseline` is the CPU oracle
(a C port of the reference's scipy path) timed on this box's host cores.

Order of the measurements (all of them are reported): the metric workload is
PREPARED first (plan, fields, output buffers), then the copy ceiling and the
`extra` workloads are measured, then -- last, with the GPU in the steady power
state those runs leave it in -- the W warm-up and K timed steps of the metric
workload.  Why: tools/clock_ramp.py and BENCH_STEP_MARKS show that an MI355X
coming out of an idle period (plan building is host work) runs the same
kernel 4-15 % slower for its first 25-30 launches (~12 ms), longer than a
5 + 20 launch run lasts; see DESIGN.md section 5.
"""
import argparse
import gc
import json
import os
import sys
import time

_REPO = os.path.dirname(os.path.abspath(__file__))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

KERNEL_OF_FAMILY = {'rowscalar': 'spmm_rowscalar', 'rowgroup': 'spmm_rowgroup',
                    'patch': 'spmm_patch'}
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='config3',
                    help='a key of pyremap_amd.synthetic.CONFIGS')
    ap.add_argument('--fields', type=int, default=None,
                    help='override the number of batched fields K')
    ap.add_argument('--mode', default='fracb',
                    choices=['fracb', 'masked', 'raw'])
    ap.add_argument('--layout', default='nk', choices=['nk', 'tnl', 'tn'],
                    help="'nk': field (n_a, K); 'tnl': (T=8, n_a, K/8); "
                         "'tn': (T=K, n_a)")
    ap.add_argument('--locality', default='mesh',
                    choices=['raster', 'mesh', 'scatter', 'none'],
                    help="numbering of the synthetic source mesh: 'mesh' = "
                         "as MPAS numbers its cells (default), 'raster' = "
                         "along the destination raster (round 2), 'scatter' "
                         "= at random")
    ap.add_argument('--shard', default='rows', choices=['rows', 'fields'])
    ap.add_argument('--sets', type=int, default=3,
                    help='distinct X/Y buffer sets rotated over the steps')
    ap.add_argument('--tune', default='',
                    help='comma-separated remap_apply_args.tune values')
    ap.add_argument('--flags', type=int, default=0)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extra', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--backend', default='nccl',
                    help="torch.distributed backend: 'nccl' (= RCCL, the "
                         "real thing) or 'gloo' to rehearse the N > 1 code "
                         "path with several ranks sharing one GPU")
    ap.add_argument('--metric-first', action='store_true',
                    help='time the metric workload BEFORE the extras (A/B '
                         'of the idle-state effect; see the module docstring)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise RCCL even with one rank (exercises '
                         'the N > 1 code path on a 1-GPU box)')
    return ap.parse_args()


def init_dist(args):
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(
                f'--gpus {args.gpus} needs one process per GPU: launch with '
                f'python -m torch.distributed.run --nproc-per-node '
                f'{args.gpus} bench.py --gpus {args.gpus}')
        raise SystemExit(f'WORLD_SIZE={world} but --gpus {args.gpus}')
    if rank != 0:
        # only rank 0 reports: nothing another rank (or the libraries it
        # loads) writes may land behind rank 0's JSON line
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if args.backend != 'nccl':
        local = local % max(torch.cuda.device_count(), 1)   # rehearsal
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(args.backend, rank=rank,
                                    world_size=world)
        # the first collective makes RCCL connect (and print its version
        # banner on stdout): have that happen here, not after the JSON line
        dist.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush()
    return rank, world, local, dist


def barrier(dist):
    import torch
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def time_steps(launch, steps, warmup, dist):
    """
    warmup untimed, then EXACTLY `steps` timed between barrier + synchronize.
    Returns (wall_s, mean_ms, [per-launch ms]).

    ``mean_ms`` is the average launch duration over the timed region: ONE pair
    of HIP events on the launch stream (torch's current stream is the stream
    handed to the C ABI) around the `steps` back-to-back launches.  An event
    pair per step costs 7.6 us per step (tools/launch_gap.py) -- 2 % of a
    full config-3 launch, 12 % of a 1/8 row shard -- so the per-launch
    spread (median / min) comes from a second, untimed pass.
    """
    import torch
    for i in range(warmup):
        launch(i)
    first = torch.cuda.Event(enable_timing=True)
    last = torch.cuda.Event(enable_timing=True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps)] \
        if os.environ.get('BENCH_STEP_MARKS') else None
    barrier(dist)
    t0 = time.perf_counter()
    first.record()
    for i in range(steps):
        launch(warmup + i)
        if marks:
            marks[i].record()     # diagnosis only: where the region's time is
    last.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if marks:
        prev = first
        gaps = []
        for ev in marks:
            gaps.append(round(prev.elapsed_time(ev), 4))
            prev = ev
        print('BENCH_STEP_MARKS', gaps, file=sys.stderr)
    if dist is not None:
        t = torch.tensor([wall], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    mean_ms = first.elapsed_time(last) / steps
    n = min(steps, 50)
    events = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        events[i][0].record()
        launch(warmup + i)
        events[i][1].record()
    torch.cuda.synchronize()
    per_launch = [a.elapsed_time(b) for a, b in events]
    return wall, mean_ms, per_launch


def graph_replay_ms(launch, calls=48, reps=5):
    """
    The same launches replayed from ONE hipGraph (torch.cuda.CUDAGraph):
    `remap_apply_f64` neither allocates nor synchronises, so it is capturable.
    Short launches -- one 2-D field: 7 us of GPU work -- are bounded by the
    ~12 us the Python call takes when issued one by one; replayed they are
    not.  ms per launch, or None where capture is not possible.
    """
    import torch
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(calls):
                launch(i)
        graph.replay()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            graph.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / (reps * calls)
    except RuntimeError:
        torch.cuda.synchronize()
        return None


def make_fields(n_a, K, layout, sets, seed, device, nan_frac=0.0,
                dtype='f64', times=8):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = []
    for _ in range(sets):
        if layout == 'nk':
            shape, cell_axis = (n_a, K), 0
        elif layout == 'tn':
            shape, cell_axis = (K, n_a), 1
        else:
            shape, cell_axis = (times, n_a, K // times), 1
        x = torch.randn(shape, generator=g, device=device,
                        dtype=torch.float64)
        if dtype == 'f32':
            x = x.to(torch.float32)
        if nan_frac:
            # whole source cells missing in every field (land / ice shelf)
            dead = torch.rand(n_a, generator=g, device=device) < nan_frac
            x.index_fill_(cell_axis, dead.nonzero().squeeze(1), float('nan'))
        out.append(x)
    return out


class Workload:
    """One prepared workload: plan, resident fields, output buffers."""


def prepare(name, args, rank, world, dist, K=None, mode=None, layout=None,
            sets=None, dtype='f64', locality=None, times=8):
    """Build plan + fields + output buffers for one workload."""
    import torch

    from pyremap_amd import engine, synthetic
    device = torch.device('cuda', torch.cuda.current_device())
    w = Workload()
    cfg = synthetic.CONFIGS[name]
    w.name, w.title = name, cfg['title']
    w.K = K or args.fields or cfg['K']
    w.mode = mode or args.mode
    w.layout = layout or args.layout
    w.sets = sets or args.sets
    w.emode = {'fracb': engine.MODE_FRACB, 'masked': engine.MODE_MASKED,
               'raw': engine.MODE_RAW}[w.mode]
    tune = [int(t) for t in args.tune.split(',')] if args.tune else None

    t0 = time.perf_counter()
    w.locality = locality or args.locality
    m = synthetic.make_config(name, device=device, locality=w.locality)
    full = engine.RemapPlan.from_triplets(
        m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, index_base=1,
        device=device)
    w.m, w.full = m, full
    w.K_local = w.K
    w.plan = full
    w.sharded = dist is not None and args.shard == 'rows'
    w.remap = None
    if w.sharded:
        from pyremap_amd.parallel import ShardedRemap
        w.remap = ShardedRemap(full, grid_dims=None)
        w.plan = w.remap.plan
    elif dist is not None:
        w.K_local = w.K // world
    # what Remapper does after loading a mapping: pick the schedule for the
    # rows this rank owns
    if not args.tune:
        w.schedule = w.plan.auto_schedule(m.dst_dims)
    else:
        w.schedule = {'family': 'explicit tune', 'tune': args.tune}
    torch.cuda.synchronize()
    w.plan_s = time.perf_counter() - t0
    w.dtype = dtype
    w.times = times
    w.fields = make_fields(m.n_a, w.K_local, w.layout, w.sets, 1234, device,
                           nan_frac=0.25 if w.mode == 'masked' else 0.0,
                           dtype=dtype, times=times)
    w.exchange = None
    w.full_field = None
    if w.sharded:
        w.exchange = time_exchange(w, dist)

    dst = None if w.plan.n_b != w.plan.n_b_global else m.dst_dims
    w.outs = [None] * w.sets
    axes = [0] if w.layout == 'nk' else [1]

    def launch(i):
        s = i % w.sets
        w.outs[s] = engine.remap_tensor(
            w.plan, dst, w.fields[s], axes, w.emode, threshold=0.01,
            flags=args.flags, tune=tune, out=w.outs[s])

    w.launch = launch
    # every output buffer exists before anything else is allocated (buffers
    # that recycled the copy-ceiling's freed blocks once measured 4 % slower
    # for the whole process: placement in HBM)
    for s in range(w.sets):
        launch(s)
    torch.cuda.synchronize()
    return w


def time_exchange(w, dist):
    """
    The ONE exchange step of the sharded path in its collective form -- one
    RCCL broadcast of the whole field, then each rank gathers the packed
    source rows its shard references -- timed on its own; leaves every rank
    holding ITS packed rows of rank 0's fields.  (The form that sends each
    rank only its packed rows: time_packed.)
    """
    import torch

    from pyremap_amd import engine
    out = {}
    x = w.fields[0]
    axis = 0 if w.layout == 'nk' else 1
    times = []
    for _ in range(3):
        barrier(dist)
        t0 = time.perf_counter()
        dist.broadcast(x, src=0)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    out['broadcast_ms'] = min(times)
    for x in w.fields:            # every set's packed rows resident
        dist.broadcast(x, src=0)
    torch.cuda.synchronize()
    out['field_bytes'] = x.numel() * x.element_size()
    out['packed_fraction_of_broadcast'] = w.remap.packed_fraction()
    out['packed_rows_this_rank'] = int(w.remap.ucols.shape[0])
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    packed = [engine.gather_rows(x, axis, w.remap.ucols) for x in w.fields]
    b.record()
    torch.cuda.synchronize()
    out['local_gather_ms'] = a.elapsed_time(b) / len(w.fields)
    w.full_field = w.fields[0]     # (rank 0's is the data; time_packed)
    w.fields = packed
    return out


def time_packed(w, dist):
    """
    The exchange that moves only what is needed: rank 0 gathers each rank's
    packed source rows and ONE all_to_all_single delivers them.  Timed AFTER
    the metric, under the watchdog (see main): it has run under gloo and with
    ranks sharing one GPU, not yet across xGMI.
    """
    import torch
    if w.full_field is None or dist.get_backend() != 'nccl':
        return {}      # (gloo moves GPU tensors through the host)
    x = w.full_field
    axis = 0 if w.layout == 'nk' else 1
    times = []
    try:
        for _ in range(3):
            barrier(dist)
            t0 = time.perf_counter()
            w.remap.distribute(x, src=0, axis=axis, how='alltoall')
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        return {'packed_ms': min(times)}
    except RuntimeError as exc:
        return {'packed_ms': None,
                'packed_error': str(exc).splitlines()[0][:200]}


def time_pipelined(w, args, dist, reps=5, n_batches=4):
    """
    The whole sharded job WITH its exchange: the K fields as `n_batches`
    column batches; batch b + 1 travels (bands, or one broadcast) while batch
    b is computed.  Milliseconds per K fields, max over ranks.
    """
    import torch
    if w.remap is None or w.layout != 'nk' or w.full_field is None:
        return None
    kb = w.K_local // n_batches
    src = w.full_field
    batches = [src[:, b * kb:(b + 1) * kb].contiguous()
               for b in range(n_batches)]
    outs = [torch.empty((w.plan.n_b, kb), dtype=torch.float64,
                        device=src.device) for _ in range(n_batches)]
    out = {}
    for how in ('alltoall', 'broadcast'):
        if how == 'alltoall' and dist.get_backend() != 'nccl':
            continue
        try:
            w.remap.apply_pipelined(batches, w.emode, how=how,
                                    threshold=0.01, flags=args.flags,
                                    outs=outs)
        except RuntimeError as exc:   # gloo rehearsal: no GPU send/recv
            out[f'pipelined_{how}_error'] = str(exc).splitlines()[0][:200]
            continue
        times = []
        for _ in range(reps):
            barrier(dist)
            t0 = time.perf_counter()
            w.remap.apply_pipelined(batches, w.emode, how=how,
                                    threshold=0.01, flags=args.flags,
                                    outs=outs)
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], device='cuda',
                             dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            times.append(float(t.item()) * 1e3)
        out[f'pipelined_{how}_ms_per_K_fields'] = min(times)
    out['n_column_batches'] = n_batches
    return out


def measure(w, args, dist, steps=None, warmup=None):
    """W warm-up + K timed steps of a prepared workload."""
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    wall, mean_ms, per_launch = time_steps(w.launch, steps, warmup, dist)
    in_order = [round(t, 4) for t in per_launch[:20]]
    per_launch.sort()
    m, plan = w.m, w.plan
    # algorithmic bytes of ONE launch on this rank; only
    # source rows some entry references count towards X
    bytes_alg = plan.algorithmic_bytes(w.K_local,
                                       4 if w.dtype == 'f32' else 8, w.emode)
    return dict(
        name=w.name, title=w.title, n_a=m.n_a, n_b=m.n_b,
        n_s_file=m.n_s, nnz_csr=w.full.nnz, K=w.K, mode=w.mode,
        layout=w.layout, locality=w.locality, steps=steps, warmup=warmup,
        wall_s=wall,
        ms_per_step=wall * 1e3 / steps,
        kernel_ms_mean=mean_ms,
        kernel_ms_graph_replay=None,    # (replay_short_extras)
        kernel_ms_median=per_launch[len(per_launch) // 2],
        kernel_ms_min=per_launch[0], kernel_ms_max=per_launch[-1],
        kernel_ms_second_pass_in_order=in_order,
        touched_frac=plan.touched_sources() / max(plan.n_a, 1),
        cell_fields_per_s=m.n_b * w.K * steps / wall,
        dst_cells_per_s_per_batch=m.n_b * steps / wall,
        bytes_alg=bytes_alg,
        bytes_alg_read=bytes_alg - plan.n_b * w.K_local * 8,
        achieved_GBps=bytes_alg / (mean_ms * 1e-3) / 1e9,
        plan_build_s=w.plan_s, exchange=w.exchange,
        rows_this_rank=plan.n_b, nnz_this_rank=plan.nnz,
        schedule=w.schedule,
    )


def copy_ceiling(device, reps=60):
    """The box's achievable HBM rate: 1 GiB device copies, read + write."""
    import torch

    from pyremap_amd import engine
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    src.random_(0, 255)
    for _ in range(3):
        engine.stream_copy(dst, src)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        engine.stream_copy(dst, src)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    assert torch.equal(src[:4096], dst[:4096])
    return 2 * n / (ms * 1e-3) / 1e9


def cpu_baseline(full, m, field, mode, budget_s):
    """
    The CPU oracle (C port of the reference's scipy path: sequential
    csr_matvecs + normalisation) on this box's host, same triplets, same
    field.  The sample is the WHOLE workload repeated while it fits the time
    budget; scipy's own `csr @ X` (what the reference executes) is timed
    beside it when scipy is importable.
    """
    import numpy as np

    from oracle import oracle
    rowptr, col, val = full.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (full.n_b, full.n_a))
    frac_b = full.frac_b.cpu().numpy()
    X = field.cpu().numpy().reshape(full.n_a, -1)
    K = X.shape[1]
    masked = mode == 'masked'

    def timed(nthreads, budget):
        times = []
        t_end = time.perf_counter() + budget
        while len(times) < 5 and (not times or time.perf_counter() < t_end):
            t0 = time.perf_counter()
            oracle.remap_flat(csr, frac_b, X, masked, 0.01, nthreads=nthreads)
            times.append(time.perf_counter() - t0)
        return min(times), len(times)

    t1, reps1 = timed(1, budget_s * 0.6)
    ncores = os.cpu_count() or 1
    nthr = min(ncores, oracle.load().oracle_max_threads())
    tn, repsn = timed(nthr, budget_s * 0.2)
    out = dict(
        value=full.n_b * K / t1, unit='dst cell-fields/s', cores=1,
        kind='port',
        sample=f'whole workload ({full.n_a} -> {full.n_b} cells, K = {K}, '
               f'mode {mode}), best of {reps1} runs of the C oracle on 1 '
               f'thread',
        seconds=t1,
        all_cores=dict(value=full.n_b * K / tn, cores=nthr, seconds=tn,
                       runs=repsn),
        host_cpus=ncores,
    )
    try:
        import scipy.sparse as sp
        A = sp.csr_matrix((val, col, rowptr), shape=(full.n_b, full.n_a))
        Xs = np.nan_to_num(X) if masked else X
        t0 = time.perf_counter()
        A.dot(Xs)
        ts = time.perf_counter() - t0
        out['scipy_spmm_only'] = dict(
            value=full.n_b * K / ts, seconds=ts, cores=1,
            note='bare scipy csr @ X (remap_numpy.py:268), no '
                 'normalisation/mask passes')
    except ImportError:
        pass
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    out['host_model'] = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return out


def pcie_inclusive(args):
    """
    The same workload when the boundary hands over HOST buffers (numpy in,
    numpy out through Remapper.remap_array): upload of X, launch, download
    of Y and of the byte mask.  Reported for DESIGN.md; never `value`.
    Two layouts: the (n_a, K) matrix of the metric and the layout MPAS output
    has, (Time, nCells, nVertLevels), which pipelines batch by batch.
    """
    import numpy as np
    import torch

    from pyremap_amd import Remapper, synthetic
    cfg = synthetic.CONFIGS['config3']
    m = synthetic.make_config('config3', device='cuda',
                              locality=args.locality)

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
    out = {}
    for tag, shape, axes in (('n_a_K', (m.n_a, cfg['K']), [0]),
                             ('Time8_nCells_L64', (8, m.n_a, cfg['K'] // 8),
                              [1])):
        x = rng.standard_normal(shape)
        r.remap_array(x[..., :8], axes)              # loads the weights
        times = []
        y = None
        for _ in range(4):
            del y
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = r.remap_array(x, axes)
            times.append(time.perf_counter() - t0)
        t = min(times)
        out[tag] = dict(
            seconds=t, seconds_first_call=times[0],
            cell_fields_per_s=m.n_b * cfg['K'] / t,
            bytes_over_pcie=x.nbytes + y.data.nbytes + y.mask.nbytes)
        del x, y
    out['numbering'] = args.locality
    out['note'] = ('numpy in -> numpy masked array out through '
                   'Remapper.remap_array, best of 4')
    return out


def load_traffic(name, K, mode, locality='mesh'):
    """PMC-measured HBM bytes per launch, from a committed rocprofv3 run
    (`traffic_<workload>_mesh.json` for the mesh numbering, the round-2
    files `traffic_<workload>.json` for the raster numbering)."""
    suffix = {'raster': '', 'mesh': '_mesh'}.get(locality)
    if suffix is None:
        return None, None
    if mode != 'fracb':
        suffix += '_' + mode
    path = os.path.join(_REPO, 'profiles', f'traffic_{name}{suffix}.json')
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        t = json.load(f)
    if t.get('K') != K or t.get('mode') != mode:
        return None, None
    return t.get('hbm_bytes_per_launch'), t.get('source')


EXTRA_KEYS = ('title', 'n_a', 'n_b', 'nnz_csr', 'K', 'mode', 'layout',
              'locality', 'schedule', 'touched_frac', 'ms_per_step',
              'kernel_ms_mean', 'kernel_ms_graph_replay', 'kernel_ms_median',
              'cell_fields_per_s',
              'bytes_alg', 'achieved_GBps')

#: workloads that hold tens of GB: prepared, measured and freed one at a
#: time, before the small ones are prepared
BIG = ('headline', 'config4', 'config5')


def extras_todo(args, world):
    """
    The other reported workloads: (tag, prepare() keywords, steps).  Their
    summary rows land in `roofline.workloads` (the driver's record keeps
    `roofline`; it truncates `extra`).
    """
    if args.no_extra or args.workload != 'config3':
        return []
    if world == 1:
        return [
            # north_star's target workload; BASELINE configs 5 and 4
            ('headline', dict(name='headline', sets=2), 12),
            ('config5', dict(name='config5', sets=1), 4),
            ('config5_masked', dict(name='config5', sets=1, mode='masked'),
             4),
            ('config4', dict(name='config4', sets=1), 6),
            ('config4_f32_fields', dict(name='config4', sets=1, dtype='f32'),
             6),
            # the metric mapping in the raster numbering round 2 measured
            ('config3_raster_numbering',
             dict(name='config3', locality='raster'), 50),
            ('K1_one_2d_field', dict(name='config3', K=1), 50),
            ('K12_monthly_time_nCells', dict(name='config3', K=12,
                                             layout='tn'), 50),
            ('Time120_nCells', dict(name='config3', K=120, layout='tn'), 30),
            # lat-lon model output on BASELINE config 1's bilinear map:
            # (time = 120, lat x lon) float32, the reference's real input
            ('config1_time120_latlon_f32',
             dict(name='config1', K=120, layout='tn', dtype='f32'), 30),
            # the same map as ESMF makes it: pole-cap rows of 360 entries,
            # a third of all entries -- applied apart (long rows)
            ('config1_esmf_pole_caps_K1', dict(name='config1_esmf', K=1), 50),
            ('config1_esmf_pole_caps_K64', dict(name='config1_esmf', K=64),
             50),
            ('f32_fields', dict(name='config3', dtype='f32'), 50),
            ('layout_T8_nCells_L64', dict(name='config3', layout='tnl'), 50),
            ('layout_T8_nCells_L60', dict(name='config3', layout='tnl',
                                          K=480), 50),
            # short level runs (10 soil / ice layers): small LDS patches
            ('layout_T48_nCells_L10', dict(name='config3', layout='tnl',
                                           K=480, times=48), 30),
            ('masked', dict(name='config3', mode='masked'), 50),
        ]
    return [('masked', dict(name='config3', mode='masked'), 50)]


def prepare_extras(args, rank, world, dist, extra, big):
    """Build the extra workloads of one class (host work: the GPU idles)."""
    ready = []
    for tag, kw, steps in extras_todo(args, world):
        if (kw['name'] in BIG) != big:
            continue
        kw = dict(kw)
        try:
            ready.append((tag, prepare(kw.pop('name'), args, rank, world,
                                       dist, **kw), steps))
        except Exception as exc:  # noqa: BLE001 - report, keep the line
            extra[tag] = {'error': f'{type(exc).__name__}: {exc}'}
    return ready


def measure_big_extras(args, rank, world, dist, extra):
    """The tens-of-GB workloads: prepare, measure, free -- one at a time."""
    import torch
    for tag, kw, steps in extras_todo(args, world):
        if kw['name'] not in BIG:
            continue
        kw = dict(kw)
        try:
            w = prepare(kw.pop('name'), args, rank, world, dist, **kw)
            measure_extras([(tag, w, steps)], args, dist, extra,
                           long_last=False)
            w.launch = w.fields = w.outs = None   # (launch closes over w)
            del w
        except Exception as exc:  # noqa: BLE001
            extra[tag] = {'error': f'{type(exc).__name__}: {exc}'}
        gc.collect()
        torch.cuda.empty_cache()


def measure_extras(ready, args, dist, extra, long_last=True):
    """
    Time the prepared extras back to back (no host work in between).  The
    last one -- the metric mapping in masked mode, same kernel family -- runs
    for at least ~30 ms of GPU time (more steps when the kernel is short, e.g.
    a 1/8 row shard), so the GPU is in its busy power state when the metric
    workload's warm-up starts.
    """
    import torch
    for n, (tag, w, steps) in enumerate(ready):
        try:
            if long_last and n + 1 == len(ready):
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(3):
                    w.launch(i)
                b.record()
                torch.cuda.synchronize()
                est_ms = max(a.elapsed_time(b) / 3, 1e-3)
                steps = max(steps, min(3000, int(30.0 / est_ms)))
                if dist is not None:     # the same count on every rank
                    t = torch.tensor([steps], device='cuda',
                                     dtype=torch.int64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    steps = int(t.item())
            r = measure(w, args, dist, steps=steps,
                        warmup=2 if w.name in BIG else 5)
            extra[tag] = {k: r[k] for k in EXTRA_KEYS}
            extra[tag]['steps'] = steps
            extra[tag]['dtype'] = w.dtype
            extra[tag]['times'] = getattr(w, 'times', 8)
            extra[tag]['frac_of_peak'] = r['achieved_GBps'] / HBM_PEAK_GBPS
            extra[tag]['read_frac_of_peak'] = r['bytes_alg_read'] / (
                r['kernel_ms_mean'] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            # (the committed counter traffic is that of the (n_a, K) layout)
            traffic, _ = load_traffic(w.name, w.K, w.mode, w.locality) \
                if w.layout == 'nk' and w.dtype == 'f64' else (None, None)
            extra[tag]['traffic'] = traffic
        except Exception as exc:  # noqa: BLE001
            extra[tag] = {'error': f'{type(exc).__name__}: {exc}'}


def replay_short_extras(ready, extra):
    """Launches the host cannot issue as fast as the GPU finishes them
    (< 50 us), replayed from a hipGraph -- after the metric is in hand."""
    for tag, w, _ in ready:
        e = extra.get(tag)
        if isinstance(e, dict) and e.get('kernel_ms_mean', 1.0) < 0.05:
            e['kernel_ms_graph_replay'] = graph_replay_ms(w.launch)


def workload_rows(extra):
    """`roofline.workloads`: one short row per extra workload."""
    rows = {}
    for tag, e in extra.items():
        if not isinstance(e, dict) or 'kernel_ms_mean' not in e:
            if isinstance(e, dict) and 'error' in e:
                rows[tag] = {'error': e['error'][:120]}
            continue
        sched = e.get('schedule') or {}
        rows[tag] = {
            'ms': round(e['kernel_ms_mean'], 5),
            'frac': round(e['frac_of_peak'], 4),
            # launched one by one from Python / replayed from one hipGraph
            # (short launches only: the host call takes ~12 us)
            'ms_graph_replay': (round(e['kernel_ms_graph_replay'], 5)
                                if e.get('kernel_ms_graph_replay') else None),
            'frac_graph_replay': (round(
                e['bytes_alg'] / (e['kernel_ms_graph_replay'] * 1e-3) / 1e9 /
                HBM_PEAK_GBPS, 4) if e.get('kernel_ms_graph_replay')
                else None),
            'read_frac': round(e['read_frac_of_peak'], 4),
            'traffic_ratio': (round(e['traffic'] / e['bytes_alg'], 4)
                              if e.get('traffic') else None),
            'GBps': round(e['achieved_GBps'], 1),
            'bytes_alg': e['bytes_alg'],
            'K': e['K'], 'mode': e['mode'], 'layout': e['layout'],
            'dtype': e.get('dtype', 'f64'), 'numbering': e['locality'],
            'kernel': KERNEL_OF_FAMILY.get(sched.get('family'), 'spmm_*') +
            ' + spmm_patchcell (long rows apart)' if sched.get('long_rows')
            else 'spmm_patchcell' if e['layout'] == 'tn' else
            'spmm_patch' if e['layout'] == 'tnl' and
            4 <= e['K'] // e.get('times', 8) < 16 else
            'spmm_rowlane' if e['K'] <= 32 else
            KERNEL_OF_FAMILY.get(sched.get('family'), 'spmm_*'),
        }
    return rows


def print_line(line):
    """The JSON line is the LAST thing on stdout ... through C stdio, which a
    pipe makes fully buffered: flush it first, or RCCL's version banner lands
    behind the line at exit."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


#: seconds the optional point-to-point measurements of an N > 1 run may take
#: before the metric is reported without them
OPTIONAL_TIMEOUT_S = int(os.environ.get('BENCH_OPTIONAL_TIMEOUT_S', 150))


def compose_line(args, res, world, ceiling, cpu, extra, pipelined,
                 status='ok'):
    """The one JSON line, from what has been measured."""
    K = res['K']
    traffic, traffic_src = load_traffic(args.workload, K, res['mode'],
                                        args.locality)
    kernel_ms = res.get('kernel_ms_mean_max_rank', res['kernel_ms_mean'])
    achieved = res['bytes_alg'] / (kernel_ms * 1e-3) / 1e9
    multi = None
    if res['exchange'] is not None:
        multi = dict(res['exchange'])
        multi.update(pipelined or {})
        multi['kernel_phase_ms'] = kernel_ms
    line = {
        'metric': 'dst cell-fields/s (dst cells x batched fields per second)'
                  ' + HBM GB/s, EC30to60 MPAS -> 0.5deg lat-lon, 512 batched '
                  'fp64 fields',
        'value': res['cell_fields_per_s'],
        'unit': 'dst cell-fields/s',
        'n_gpus': world,
        'steps': res['steps'],
        'warmup': res['warmup'],
        'ms_per_step': res['ms_per_step'],
        'higher_is_better': True,
        'scaling': 'strong',   # the problem is fixed; rows (or fields) are
        #                        divided over the ranks
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': f"{args.workload}: {res['title']}",
            'n_a': res['n_a'], 'n_b': res['n_b'], 'n_s': res['n_s_file'],
            'nnz_csr': res['nnz_csr'], 'fields_K': K,
            'mode': res['mode'], 'layout': res['layout'],
            'locality': args.locality,
            'touched_frac': res['touched_frac'],
            'sharding': 'none' if world == 1 else
            (f'dst rows over {world} GPUs in packed column space, X '
             f'distributed once (RCCL) before the timed region'
             if args.shard == 'rows' else
             f'fields over {world} GPUs, no collective'),
            'buffer_sets_rotated': args.sets,
            'bitwise_mode': not (args.flags & 1),
            'schedule': res['schedule'],
            'measurement_order': 'metric first' if args.metric_first else
            'everything prepared first; then copy ceiling, extras, metric '
            'workload timed back to back',
        },
        'dst_cells_per_s_per_512_batch': res['dst_cells_per_s_per_batch'],
        'roofline': {
            'bound': 'hbm',
            'achieved': achieved,
            'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBPS,
            'traffic': traffic,
            'traffic_source': traffic_src,
            'traffic_note': 'FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 per '
            'launch (MI355X_MICROARCH.md): the L2\'s fabric-side bytes, '
            'Infinity Cache hits included -- an upper bound on HBM bytes',
            'kernel': KERNEL_OF_FAMILY.get(
                res['schedule'].get('family'), 'spmm_*') +
            ' (remap_apply_f64)',
            'kernel_ms_mean': kernel_ms,
            'kernel_ms_median': res['kernel_ms_median'],
            'kernel_ms_min': res['kernel_ms_min'],
            'kernel_ms_max': res['kernel_ms_max'],
            'kernel_ms_second_pass_in_order':
            res['kernel_ms_second_pass_in_order'],
            'kernel_ms_steady_100_more': res['kernel_ms_steady_100_more'],
            'bytes_alg_per_launch': res['bytes_alg'],
            'bytes_alg_read_per_launch': res['bytes_alg_read'],
            'read_frac_of_peak': res['bytes_alg_read'] /
            (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            'measured_copy_ceiling_GBps': ceiling,
            'numbering': args.locality,
            'workloads': workload_rows(extra),
        },
        'status': status,
        'cpu_baseline': cpu,
        'plan_build_s': res['plan_build_s'],
        'multi_gpu': multi,
        'extra': extra,
    }
    return line


def main():
    args = parse_args()
    import torch
    rank, world, local, dist = init_dist(args)
    device = torch.device('cuda', local)
    from pyremap_amd import engine
    engine.require_gpu()

    extra = {}
    # The metric workload and the small extras are PREPARED first (host
    # work, GPU mostly idle); the tens-of-GB workloads come and go one at a
    # time in between; then all remaining measurements run back to back, the
    # metric workload last.
    main_w = prepare(args.workload, args, rank, world, dist)
    ready = prepare_extras(args, rank, world, dist, extra, big=False)
    measure_big_extras(args, rank, world, dist, extra)
    ceiling = copy_ceiling(device)
    pipelined = None
    res = None
    if args.metric_first:
        res = measure(main_w, args, dist)
    measure_extras(ready, args, dist, extra)
    if res is None:
        res = measure(main_w, args, dist)
    # a long run behind it, for the record: the steady-state launch time
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(100):
        main_w.launch(i)
    b.record()
    torch.cuda.synchronize()
    res['kernel_ms_steady_100_more'] = a.elapsed_time(b) / 100
    if dist is None:
        replay_short_extras(ready, extra)
    # per-rank kernel numbers -> the slowest rank prices the roofline
    if dist is not None:
        t = torch.tensor([res['kernel_ms_mean']], device=device,
                         dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res['kernel_ms_mean_max_rank'] = float(t.item())
    watchdog = None
    if main_w.sharded:
        # The metric is in hand.  What follows moves packed rows with
        # all_to_all_single over RCCL, which no test could exercise on more
        # than one GPU: should it not return, rank 0 prints the line with
        # "status": "exchange_hung" and every rank exits with code 3 -- a
        # hung exchange must not look like a clean run.
        import threading

        def bail():
            if rank == 0:
                late = dict(res['exchange'] or {})
                late['optional_measurements'] = (
                    f'timed out after {OPTIONAL_TIMEOUT_S} s: packed / '
                    f'pipelined exchange did not return')
                res['exchange'] = late
                print_line(compose_line(args, res, world, ceiling, None,
                                        extra, None,
                                        status='exchange_hung'))
            os._exit(3)
        watchdog = threading.Timer(OPTIONAL_TIMEOUT_S, bail)
        watchdog.daemon = True
        watchdog.start()
        if os.environ.get('BENCH_TEST_HANG'):      # exercises the watchdog
            time.sleep(10 ** 6)
        packed = time_packed(main_w, dist)
        if res['exchange'] is not None:
            res['exchange'].update(packed)
        pipelined = time_pipelined(main_w, args, dist)
        barrier(dist)
        watchdog.cancel()
    del ready

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(main_w.full, main_w.m, main_w.fields[0],
                           res['mode'], args.cpu_seconds)
    if world == 1 and not args.no_extra and args.workload == 'config3':
        main_w.fields = main_w.outs = None
        torch.cuda.empty_cache()
        try:
            extra['host_buffers_pcie_inclusive'] = pcie_inclusive(args)
        except Exception as exc:  # noqa: BLE001
            extra['host_buffers_pcie_inclusive'] = {
                'error': f'{type(exc).__name__}: {exc}'}

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    line = compose_line(args, res, world, ceiling, cpu, extra, pipelined)
    # the JSON line is the LAST thing on stdout: every rank is done first
    # (RCCL prints a version banner on stdout when its communicator comes up
    # or goes down, whichever happens to be later)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    print_line(line)


if __name__ == '__main__':
    main()
