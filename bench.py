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
rows are sharded over the ranks (nnz-balanced contiguous ranges), rank 0's
source field is broadcast ONCE over RCCL/xGMI before the timed region (timed
separately, `bcast_ms`), and the timed steps contain no collective.  The
problem size is fixed, so this is strong scaling.

Prints ONE JSON line on rank 0.  `value` = destination cell-fields per second
for the whole job; `roofline` prices the kernel against HBM bandwidth using
SURVEY.md section 8(d)'s algorithmic bytes; `cpu_baseline` is the CPU oracle
(a C port of the reference's scipy path) timed on this box's host cores.
"""
import argparse
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
    ap.add_argument('--layout', default='nk', choices=['nk', 'tnl'],
                    help="'nk': field (n_a, K); 'tnl': (T=8, n_a, K/8)")
    ap.add_argument('--locality', default='raster',
                    choices=['raster', 'none'])
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
    barrier(dist)
    t0 = time.perf_counter()
    first.record()
    for i in range(steps):
        launch(warmup + i)
    last.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
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


def make_fields(n_a, K, layout, sets, seed, device, nan_frac=0.0):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = []
    for _ in range(sets):
        if layout == 'nk':
            x = torch.randn((n_a, K), generator=g, device=device,
                            dtype=torch.float64)
        else:
            x = torch.randn((8, n_a, K // 8), generator=g, device=device,
                            dtype=torch.float64)
        if nan_frac:
            # whole source cells missing in every field (land / ice shelf)
            dead = torch.rand(n_a, generator=g, device=device) < nan_frac
            if layout == 'nk':
                x[dead, :] = float('nan')
            else:
                x[:, dead, :] = float('nan')
        out.append(x)
    return out


def run_workload(name, args, rank, world, dist, K=None, mode=None,
                 layout=None, steps=None, warmup=None, sets=None):
    """Build plan + fields for one workload and time it."""
    import torch

    from pyremap_amd import engine, synthetic
    device = torch.device('cuda', torch.cuda.current_device())
    cfg = synthetic.CONFIGS[name]
    K = K or args.fields or cfg['K']
    mode = mode or args.mode
    layout = layout or args.layout
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    sets = sets or args.sets
    emode = {'fracb': engine.MODE_FRACB, 'masked': engine.MODE_MASKED,
             'raw': engine.MODE_RAW}[mode]
    tune = [int(t) for t in args.tune.split(',')] if args.tune else None

    t0 = time.perf_counter()
    m = synthetic.make_config(name, device=device, locality=args.locality)
    full = engine.RemapPlan.from_triplets(
        m.row, m.col, m.S, m.frac_b, m.n_a, m.n_b, index_base=1,
        device=device)
    K_local = K
    plan = full
    bcast_ms = None
    sharded = dist is not None and args.shard == 'rows'
    if sharded:
        plan = full.shard(rank, world)
    elif dist is not None:
        K_local = K // world
    # what Remapper does after loading a mapping: pick the schedule for the
    # rows this rank owns
    if not args.tune:
        schedule = plan.auto_schedule(m.dst_dims)
    else:
        schedule = {'family': 'explicit tune', 'tune': args.tune}
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    fields = make_fields(m.n_a, K_local, layout, sets, 1234 + 0 * rank,
                         device, nan_frac=0.25 if mode == 'masked' else 0.0)
    if sharded:
        # the ONE exchange step of the path: rank 0's fields go to all ranks
        barrier(dist)
        tb = time.perf_counter()
        for x in fields:
            dist.broadcast(x, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - tb) * 1e3 / len(fields)

    dst = None if plan.n_b != plan.n_b_global else m.dst_dims
    outs = [None] * sets

    def launch(i):
        s = i % sets
        outs[s] = engine.remap_tensor(plan, dst, fields[s],
                                      [0] if layout == 'nk' else [1], emode,
                                      threshold=0.01, flags=args.flags,
                                      tune=tune, out=outs[s])

    launch(0)
    torch.cuda.synchronize()
    wall, mean_ms, per_launch = time_steps(launch, steps, warmup, dist)
    in_order = [round(t, 4) for t in per_launch[:20]]
    per_launch.sort()
    # algorithmic bytes of ONE launch on this rank; a row
    # shard reads at most the whole of X
    bytes_alg = plan.algorithmic_bytes(K_local, 8, emode)
    res = dict(
        name=name, title=cfg['title'], n_a=m.n_a, n_b=m.n_b,
        n_s_file=m.n_s, nnz_csr=full.nnz, K=K, mode=mode, layout=layout,
        steps=steps, warmup=warmup, wall_s=wall,
        ms_per_step=wall * 1e3 / steps,
        kernel_ms_mean=mean_ms, kernel_ms_median=per_launch[len(per_launch)
                                                            // 2],
        kernel_ms_min=per_launch[0], kernel_ms_max=per_launch[-1],
        kernel_ms_second_pass_in_order=in_order,
        touched_frac=plan.touched_sources() / max(plan.n_a, 1),
        cell_fields_per_s=m.n_b * K * steps / wall,
        dst_cells_per_s_per_batch=m.n_b * steps / wall,
        bytes_alg=bytes_alg, bytes_alg_read=bytes_alg - plan.n_b * K_local * 8,
        achieved_GBps=bytes_alg / (mean_ms * 1e-3) / 1e9,
        plan_build_s=plan_s, bcast_ms=bcast_ms,
        rows_this_rank=plan.n_b, nnz_this_rank=plan.nnz, schedule=schedule,
    )
    return res, full, m, fields, outs


def copy_ceiling(device):
    """The box's achievable HBM rate: 1 GiB device copy, read + write."""
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
    reps = 10
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
    """
    import numpy as np
    import torch

    from pyremap_amd import Remapper, synthetic
    cfg = synthetic.CONFIGS['config3']
    m = synthetic.make_config('config3', device='cuda')

    class Desc:
        pass
    src, dst = Desc(), Desc()
    src.dims, src.dim_sizes = ['nCells'], [m.n_a]
    dst.dims, dst.dim_sizes = ['lat', 'lon'], list(m.dst_dims)
    dst.coords, dst.mesh_name = {}, 'bench'
    mm = m.numpy()
    r = Remapper.from_triplets(mm['row'], mm['col'], mm['S'], mm['frac_b'],
                               src, dst, device='cuda')
    x = np.random.default_rng(0).standard_normal((m.n_a, cfg['K']))
    r.remap_array(x[:, :8], [0])                     # loads the weights
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = r.remap_array(x, [0])
        times.append(time.perf_counter() - t0)
    t = min(times)
    return dict(seconds=t, cell_fields_per_s=m.n_b * cfg['K'] / t,
                bytes_over_pcie=x.nbytes + y.data.nbytes + y.mask.nbytes,
                note='numpy in -> numpy masked array out, pageable host '
                     'memory, best of 3')


def load_traffic(name, K, mode):
    """PMC-measured HBM bytes per launch, from a committed rocprofv3 run."""
    path = os.path.join(_REPO, 'profiles', f'traffic_{name}.json')
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        t = json.load(f)
    if t.get('K') != K or t.get('mode') != mode:
        return None, None
    return t.get('hbm_bytes_per_launch'), t.get('source')


def main():
    args = parse_args()
    import torch
    rank, world, local, dist = init_dist(args)
    device = torch.device('cuda', local)
    from pyremap_amd import engine
    engine.require_gpu()

    res, full, m, fields, outs = run_workload(args.workload, args, rank,
                                              world, dist)
    K = res['K']
    # per-rank kernel numbers -> the slowest rank prices the roofline
    if dist is not None:
        t = torch.tensor([res['kernel_ms_mean']], device=device,
                         dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res['kernel_ms_mean_max_rank'] = float(t.item())

    extra = {}
    cpu = None
    ceiling = None
    if rank == 0:
        ceiling = copy_ceiling(device)
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(full, m, fields[0], res['mode'], args.cpu_seconds)
    if world == 1 and not args.no_extra and args.workload == 'config3':
        del fields, outs, full
        torch.cuda.empty_cache()
        for tag, kw in (
                ('masked_renormalised', dict(name='config3', mode='masked')),
                ('layout_T8_nCells_L64', dict(name='config3', layout='tnl')),
                ('headline_3.7M_to_1.0M', dict(name='headline')),
        ):
            try:
                r, f2, m2, x2, y2 = run_workload(
                    kw.pop('name'), args, rank, world, dist, steps=50,
                    warmup=5, **kw)
                extra[tag] = {k: r[k] for k in (
                    'title', 'n_a', 'n_b', 'nnz_csr', 'K', 'mode', 'layout',
                    'schedule',
                    'ms_per_step', 'kernel_ms_mean', 'cell_fields_per_s',
                    'bytes_alg', 'achieved_GBps')}
                extra[tag]['frac_of_peak'] = r['achieved_GBps'] / \
                    HBM_PEAK_GBPS
                del f2, m2, x2, y2
                torch.cuda.empty_cache()
            except Exception as exc:  # noqa: BLE001 - report, keep the line
                extra[tag] = {'error': f'{type(exc).__name__}: {exc}'}
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

    traffic, traffic_src = load_traffic(args.workload, K, res['mode'])
    kernel_ms = res.get('kernel_ms_mean_max_rank', res['kernel_ms_mean'])
    achieved = res['bytes_alg'] / (kernel_ms * 1e-3) / 1e9
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
        'scaling': 'strong' if args.shard == 'rows' else 'weak',
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
            (f'dst rows over {world} GPUs, X broadcast once (RCCL) before '
             f'the timed region' if args.shard == 'rows' else
             f'fields over {world} GPUs, no collective'),
            'buffer_sets_rotated': args.sets,
            'bitwise_mode': not (args.flags & 1),
            'schedule': res['schedule'],
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
            'kernel': KERNEL_OF_FAMILY.get(
                res['schedule'].get('family'), 'spmm_*') +
            ' (remap_apply_f64)',
            'kernel_ms_mean': kernel_ms,
            'kernel_ms_median': res['kernel_ms_median'],
            'kernel_ms_min': res['kernel_ms_min'],
            'kernel_ms_max': res['kernel_ms_max'],
            'kernel_ms_second_pass_in_order':
            res['kernel_ms_second_pass_in_order'],
            'bytes_alg_per_launch': res['bytes_alg'],
            'bytes_alg_read_per_launch': res['bytes_alg_read'],
            'read_frac_of_peak': res['bytes_alg_read'] /
            (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            'measured_copy_ceiling_GBps': ceiling,
        },
        'cpu_baseline': cpu,
        'plan_build_s': res['plan_build_s'],
        'bcast_ms': res['bcast_ms'],
        'extra': extra,
    }
    # the JSON line is the LAST thing on stdout: every rank is done first
    # (RCCL prints a version banner on stdout when its communicator comes up
    # or goes down, whichever happens to be later)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    try:
        # ... through C stdio, which a pipe makes fully buffered: flush it,
        # or the banner lands behind the JSON line at exit
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
