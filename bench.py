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

Prints ONE JSON line on rank 0, kept under 4 KB (the driver keeps an 8 KB tail
of stdout; tests/test_host_cpu.py pins the length).  `value` = destination
cell-fields per second for the whole job; `roofline` prices the kernel against
HBM bandwidth using SURVEY.md section 8(d)'s algorithmic bytes, with one
`[ms, frac]` pair per other workload in `roofline.workloads`; `cpu_baseline`
is scipy's `csr @ X` -- the call the reference makes at remap_numpy.py:268,
BASELINE.md section 4 "Baseline A" -- on one host core of this box, with the C
oracle (the whole `_remap_numpy_array` semantics) beside it as `port_value`.
Everything else that is measured (schedules, per-launch lists, titles, the
PCIe-inclusive rates, every field of every extra workload) goes to the side
file named in the line's `details` (gpurun_out/bench_extra.json).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment
starts the N ranks itself (a child `python -m torch.distributed.run ...
bench.py`, never an exec: the parent has not touched the GPU), relays rank
0's line as its own last line and exits with the child's code.

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
    ap.add_argument('--all-workloads', action='store_true',
                    help='every reported workload (configs 4 / 5 masked / '
                         'f32 ... : a minute), not only the default rows')
    ap.add_argument('--details', default=None,
                    help='side file for everything the line does not carry '
                         '(default gpurun_out/bench_extra.json)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--backend', default='nccl',
                    help="torch.distributed backend: 'nccl' (= RCCL, the "
                         "real thing) or 'gloo' to rehearse the N > 1 code "
                         "path with several ranks sharing one GPU")
    ap.add_argument('--no-multi-rows', action='store_true',
                    help='N > 1: skip the sharded headline / config 4 / '
                         'config 5 rows of `multi_gpu.workloads`')
    ap.add_argument('--metric-first', action='store_true',
                    help='time the metric workload BEFORE the extras (A/B '
                         'of the idle-state effect; see the module docstring)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise RCCL even with one rank (exercises '
                         'the N > 1 code path on a 1-GPU box)')
    return ap.parse_args()


def self_launch(args):
    """
    `python bench.py --gpus N` (N > 1) with no launcher around it: start the N
    ranks as a CHILD process (torch.distributed.run, one rank per GPU) -- this
    parent has not touched the GPU and never does --, pass every line the
    child prints through, rank 0's JSON line last, and return its exit code.
    """
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)]
    cmd += sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    out, _ = proc.communicate()
    lines = out.splitlines()
    last = None
    for n in range(len(lines) - 1, -1, -1):
        if lines[n].startswith('{"metric"'):
            last = lines.pop(n)
            break
    for line in lines:
        print(line)
    if last is not None:
        print(last, flush=True)
    elif proc.returncode == 0:
        print('bench.py: the ranks printed no JSON line', file=sys.stderr)
        return 1
    return proc.returncode


def init_dist(args):
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
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
    Returns (wall_s, mean_ms, [per-launch ms], shader clock in MHz right
    behind the timed region).

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
    # which clock state the region ran in: one wave spinning 20 us behind it
    # -- AFTER the wall clock has stopped: the probe (a fill + the spinning
    # wave) is no remap work, and on a 1/8 row shard (~45 us a step) it was
    # 3-5 % of the time `value` is computed from
    from pyremap_amd import engine
    mhz = engine.clock_probe(torch.cuda.current_device())
    torch.cuda.synchronize()
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
    return wall, mean_ms, per_launch, mhz()


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


def make_launch(w, dst, axes, tune):
    from pyremap_amd import engine
    w.launch_args = (dst, axes, tune)

    def launch(i):
        s = i % w.sets
        w.outs[s] = engine.remap_tensor(
            w.plan, dst, w.fields[s], axes, w.emode, threshold=0.01,
            flags=w.flags, tune=tune, out=w.outs[s])
    return launch


def same_with(w, flags=None, mode=None, mask='cells', tnl=0):
    """The prepared workload `w` once more with other REMAP_FLAG_* bits or in
    the masked mode: the same plan and buffers, nothing rebuilt.  (Masked:
    a quarter of the source cells of `w`'s fields become NaN IN PLACE, as
    make_fields makes them -- measure `w` itself first; `mask='levels'`: the
    K columns read as (Time, 64 levels), every cell missing below a depth of
    its own as well -- bathymetry: validity differs from column to column.
    `tnl` = T: the buffers of an (n_a, K) workload read as a (T, nCells, K /
    T) field, MPAS's layout -- fresh values, a quarter of the cells missing
    at every time and level (land), with mask='levels' also every cell below
    a depth of its own (bathymetry: the same mask at every time).)"""
    import copy

    import torch

    from pyremap_amd import engine
    v = copy.copy(w)
    if flags is not None:
        v.flags = flags
    if tnl:
        assert w.layout in ('nk', 'tnl') and mode == 'masked'
        v.mode, v.emode = 'masked', engine.MODE_MASKED
        v.layout, v.times = 'tnl', tnl
        n_a, n_b, L = w.m.n_a, w.plan.n_b, w.K_local // tnl
        g = torch.Generator(device=w.fields[0].device)
        g.manual_seed(4321)
        dst, _, tune = w.launch_args
        dshape = [n_b] if dst is None else [int(d) for d in dst]
        v.fields = [x.view(tnl, n_a, L) for x in w.fields]
        v.outs = [y.view([tnl] + dshape + [L]) for y in w.outs]
        for x in v.fields:
            x.normal_(generator=g)
            dead = torch.rand(n_a, generator=g, device=x.device) < 0.25
            x.index_fill_(1, dead.nonzero().squeeze(1), float('nan'))
            if mask == 'levels':
                depth = torch.randint(8, L + 1, (n_a, 1), generator=g,
                                      device=x.device)
                lev = torch.arange(L, device=x.device)[None]
                x.masked_fill_((lev >= depth)[None], float('nan'))
        v.launch = make_launch(v, dst, [1], tune)
        return v
    if mode == 'masked':
        v.mode, v.emode = 'masked', engine.MODE_MASKED
        g = torch.Generator(device=w.fields[0].device)
        g.manual_seed(4321)
        axis = 0 if w.layout == 'nk' else 1
        for x in w.fields:
            if w.mode != 'masked':
                dead = torch.rand(x.shape[axis], generator=g,
                                  device=x.device) < 0.25
                x.index_fill_(axis, dead.nonzero().squeeze(1), float('nan'))
            if mask == 'levels' and w.layout == 'nk':
                depth = torch.randint(8, 65, (x.shape[0], 1), generator=g,
                                      device=x.device)
                lev = (torch.arange(x.shape[1], device=x.device) % 64)[None]
                x.masked_fill_(lev >= depth, float('nan'))
    v.launch = make_launch(v, *w.launch_args)
    if mode == 'masked' and mask == 'levels' and w.layout == 'nk' and \
            (v.flags & 32):
        # the K columns of a cell are (Time, 64 levels): told so -- K / 64
        # batches of 64 levels, 64 elements apart -- the launch can keep one
        # normaliser per (row, level) for four time slices
        # (REMAP_FLAG_BATCH_MASKS); the bytes are where they were
        K, tune = w.K_local, w.launch_args[2]

        def launch(i):
            s = i % v.sets
            engine.apply_strided(
                v.plan, v.fields[s], v.outs[s], n_batch=K // 64, k_inner=64,
                x_row_stride=K, x_batch_stride=64, y_row_stride=K,
                y_batch_stride=64, mode=v.emode, threshold=0.01,
                flags=v.flags, tune=tune)
        v.launch = launch
    return v


def prepare(name, args, rank, world, dist, K=None, mode=None, layout=None,
            sets=None, dtype='f64', locality=None, times=8, flags=None):
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
    w.flags = args.flags if flags is None else flags
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

    w.launch = make_launch(w, dst, axes, tune)
    # every output buffer exists before anything else is allocated (buffers
    # that recycled the copy-ceiling's freed blocks once measured 4 % slower
    # for the whole process: placement in HBM)
    for s in range(w.sets):
        w.launch(s)
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
    # (tens of GB: once -- every rank's synthetic fields are the same bytes
    # already, the broadcast is timed, not needed)
    piece = x
    if w.name in BIG and dist.get_backend() != 'nccl':
        # the gloo rehearsal stages GPU tensors through the host: a 256 MB
        # piece of the tens of GB exercises the call (the ranks' synthetic
        # fields are the same bytes already)
        piece = x.view(-1)[:1 << 25]
        out['broadcast_note'] = 'gloo rehearsal: 256 MB piece only'
    for _ in range(1 if w.name in BIG else 3):
        barrier(dist)
        t0 = time.perf_counter()
        dist.broadcast(piece, src=0)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    out['broadcast_ms'] = min(times)
    # what the communicator itself reports (RCCL under 'nccl')
    out['ranks'] = dist.get_world_size()
    out['backend'] = dist.get_backend()
    out['exchange'] = 'one broadcast of X + local gather of the packed ' \
        'rows, before the timed region'
    for x in w.fields[1:]:        # every set's packed rows resident
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
    wall, mean_ms, per_launch, mhz = time_steps(w.launch, steps, warmup,
                                                dist)
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
        kernel_ms_mean=mean_ms, clock_mhz=mhz,
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
    The reference's CPU path on this box's host cores, same triplets, same
    field, the WHOLE workload per run.

    `value` (Baseline A): scipy's `csr @ X` on ONE
    core -- the call the reference makes at remap_numpy.py:264-268; its
    csr_matvecs kernel is single-threaded -- best of <= 5 runs, `kind`
    "reference".  Beside it (Baseline B) the C oracle, which also does the
    normalisation and mask passes of `_remap_numpy_array`: `port_value` on
    one thread, `port_all_cores` on every core.  Where scipy does not import
    the oracle's figure is `value` and `kind` says "port".
    """
    import numpy as np

    from oracle import oracle
    rowptr, col, val = full.to_host_csr()
    csr = oracle.OracleCSR(rowptr, col, val, (full.n_b, full.n_a))
    frac_b = full.frac_b.cpu().numpy()
    X = field.cpu().numpy().reshape(full.n_a, -1)
    K = X.shape[1]
    masked = mode == 'masked'
    units = full.n_b * K

    def best_of(fn, budget, most=5):
        times = []
        t_end = time.perf_counter() + budget
        while len(times) < most and (not times or
                                     time.perf_counter() < t_end):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return min(times), len(times)

    t1, reps1 = best_of(lambda: oracle.remap_flat(
        csr, frac_b, X, masked, 0.01, nthreads=1), budget_s * 0.4)
    ncores = os.cpu_count() or 1
    nthr = min(ncores, oracle.load().oracle_max_threads())
    tn, _ = best_of(lambda: oracle.remap_flat(
        csr, frac_b, X, masked, 0.01, nthreads=nthr), budget_s * 0.2)
    what = (f'whole workload ({full.n_a} -> {full.n_b} cells, K = {K}, '
            f'mode {mode})')
    out = dict(
        value=units / t1, unit='dst cell-fields/s', cores=1, kind='port',
        sample=f'{what}: C oracle of _remap_numpy_array, 1 thread, best of '
               f'{reps1}',
        seconds=t1, port_value=units / t1, port_seconds=t1,
        port_all_cores=dict(value=units / tn, cores=nthr, seconds=tn),
        host_cpus=ncores)
    try:
        import scipy.sparse as sp
        A = sp.csr_matrix((val, col, rowptr), shape=(full.n_b, full.n_a))
        Xs = np.nan_to_num(X) if masked else X
        ts, reps = best_of(lambda: A.dot(Xs), budget_s * 0.4)
        out.update(
            value=units / ts, seconds=ts, kind='reference',
            scipy_value=units / ts,
            sample=f'{what}: scipy csr @ X (remap_numpy.py:268, Baseline A)'
                   f', 1 core, best of {reps}; port_value = the C oracle of '
                   f'the whole _remap_numpy_array, 1 thread')
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


def pcie_row(main_w):
    """
    The metric workload as every `remap_numpy(ds)` / `ncremap()` caller meets
    it: the (n_a, K) field in PAGEABLE host memory (numpy), the result back
    in host memory (remap_numpy.py:254-256, 268: the reference's whole path
    is host-side).  Upload, launch and download in column panels, both PCIe
    directions busy (host_path._panel_pipeline).  `[ms, GB/s]`: best of 4
    calls, bytes over PCIe (X up + Y down) per second.  Never `value`.
    """
    import numpy as np
    import torch

    from pyremap_amd import host_path
    x = main_w.fields[0].cpu().numpy()
    dst = main_w.m.dst_dims
    emode = 'fracb' if main_w.mode == 'fracb' else 'masked'
    times = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = host_path.remap_host_array(
            main_w.plan, dst, x, [0], mode=emode,
            threshold=None if emode == 'fracb' else 0.01).result()
        times.append(time.perf_counter() - t0)
        nbytes = x.nbytes + y.nbytes
        del y
    t = min(times[1:])
    return dict(ms=t * 1e3, GBps=nbytes / t / 1e9, first_call_ms=times[0] *
                1e3, bytes_over_pcie=nbytes, layout=f'{tuple(x.shape)} '
                f'{x.dtype} pageable numpy in -> float64 numpy out',
                form='column panels (host_path._panel_pipeline)')


def load_traffic(name, K, mode, locality='mesh', family=None):
    """PMC-measured HBM bytes per launch, from a committed rocprofv3 run
    (`traffic_<workload>_mesh.json` for the mesh numbering, the round-2
    files `traffic_<workload>.json` for the raster numbering).  The file is
    a CONSTANT of the library it was measured on: it carries that library's
    ABI version and the kernel's symbol, and is refused (traffic: null) when
    either differs from what runs now -- a later kernel must not go on
    quoting an earlier kernel's counters."""
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
    from pyremap_amd import engine
    if t.get('abi_version') != engine.ABI_VERSION:
        return None, None
    kernel = KERNEL_OF_FAMILY.get(family)
    if kernel is not None and kernel not in (t.get('kernel') or ''):
        return None, None
    return t.get('hbm_bytes_per_launch'), t.get('source')


EXTRA_KEYS = ('title', 'n_a', 'n_b', 'nnz_csr', 'K', 'mode', 'layout',
              'locality', 'schedule', 'touched_frac', 'ms_per_step',
              'kernel_ms_mean', 'clock_mhz', 'kernel_ms_graph_replay', 'kernel_ms_median',
              'cell_fields_per_s',
              'bytes_alg', 'achieved_GBps')

#: workloads that hold tens of GB: prepared, measured and freed one at a
#: time, before the small ones are prepared
BIG = ('headline', 'config4', 'config5')


#: the rows of `roofline.workloads` the DEFAULT run carries (the driver's
#: command has to stay short): north_star's target workload, BASELINE config
#: 5 in the bitwise and in the FMA mode, and the weak spots VERDICT.md names
DEFAULT_ROWS = ('headline', 'config5', 'config5_fma', 'config5_masked',
                'config5_masked_levels', 'config5_tnl_masked',
                'config5_tnl_masked_levels', 'config4', 'config2',
                'Time120_nCells', 'layout_T8_nCells_L60',
                'layout_T48_nCells_L10', 'layout_T120_nCells_L4',
                'config1_esmf_pole_caps_K1',
                'config1_esmf_pole_caps_K64', 'masked')

#: N > 1: the sharded workloads BASELINE.json names for 8 GPUs (configs 4 and
#: 5) and north_star's headline, one at a time (prepare -> exchange -> measure
#: -> free); their rows land in `multi_gpu.workloads`
MULTI_ROWS = ('headline', 'config4', 'config5')


def extras_todo(args, world):
    """
    The other reported workloads: (tag, prepare() keywords, steps).  Their
    `[ms, frac]` pairs land in `roofline.workloads`, everything else about
    them in the side file.  DEFAULT_ROWS unless --all-workloads.
    """
    if args.no_extra or args.workload != 'config3':
        return []
    if world > 1:
        rows = [('masked', dict(name='config3', mode='masked'), 50)]
        if args.shard == 'rows' and not args.no_multi_rows:
            rows = [('headline', dict(name='headline', sets=1), 12),
                    ('config4', dict(name='config4', sets=1), 6),
                    ('config5', dict(name='config5', sets=1), 4)] + rows
        return rows
    rows = [
        # north_star's target workload; BASELINE configs 5 and 4
        ('headline', dict(name='headline', sets=2), 12),
        ('config5', dict(name='config5', sets=1), 4),
        # the opt-in fused multiply-add mode (rtol 1e-13, not bitwise:
        # tests/test_gpu_parity.py::test_fma_flag_is_close_not_identical)
        ('config5_fma', dict(name='config5', sets=1, flags=1, share='config5'),
         4),
        # masked mode, a quarter of the source cells missing in every field
        # (land / ice shelf): REMAP_FLAG_CELL_MASKS (16), the form
        # `remap_tensor_auto_mode` picks when the scan finds the NaNs in
        # whole cells (one normaliser per row through the shared LDS ring:
        # csrc/spmm_cellshare.h) ...
        ('config5_masked', dict(name='config5', sets=1, mode='masked',
                                share='config5', flags=16), 4),
        # ... the same as MPAS lays a 3-D variable out, (Time = 16, nCells,
        # 64 levels), land cells missing: what the layout-aware scan
        # (remap_scan_nan_layout) picks there, REMAP_FLAG_CELL_MASKS again
        # ... and cells missing below a depth of their own as well (the K
        # columns of a cell are Time x 64 levels; bathymetry: the same mask
        # at every time): REMAP_FLAG_BATCH_MASKS (32), one normaliser per
        # (row, level) for four time slices -- what the layout-aware scan
        # (remap_scan_nan_layout) picks there
        ('config5_masked_levels', dict(name='config5', sets=1, mode='masked',
                                       share='config5', mask='levels',
                                       flags=32), 4),
        # ... the same two masks as MPAS lays a 3-D variable out, (Time = 16,
        # nCells, 64 levels): a source cell's values are 16 runs of 512 bytes
        # 1.9 GB apart, not one of 8 KiB
        ('config5_tnl_masked', dict(name='config5', sets=1, mode='masked',
                                    share='config5', flags=16, tnl=16), 4),
        ('config5_tnl_masked_levels',
         dict(name='config5', sets=1, mode='masked', share='config5',
              mask='levels', flags=32, tnl=16), 4),
        ('config4', dict(name='config4', sets=1), 6),
        # BASELINE config 2: one round of workgroups, microseconds
        ('config2', dict(name='config2'), 50),
        ('config4_f32_fields', dict(name='config4', sets=1, dtype='f32'), 6),
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
        ('config1_esmf_pole_caps_K64', dict(name='config1_esmf', K=64), 50),
        ('f32_fields', dict(name='config3', dtype='f32'), 50),
        ('layout_T8_nCells_L64', dict(name='config3', layout='tnl'), 50),
        ('layout_T8_nCells_L60', dict(name='config3', layout='tnl',
                                      K=480), 50),
        # short level runs (10 soil / ice layers): small LDS patches
        ('layout_T48_nCells_L10', dict(name='config3', layout='tnl',
                                       K=480, times=48), 30),
        # ... 4 levels: the batch-at-a-time lanes-across-rows kernel
        ('layout_T120_nCells_L4', dict(name='config3', layout='tnl',
                                       K=480, times=120), 30),
        ('masked', dict(name='config3', mode='masked'), 50),
    ]
    if not args.all_workloads:
        rows = [r for r in rows if r[0] in DEFAULT_ROWS]
    return rows


def prepare_extras(args, rank, world, dist, extra, big):
    """Build the extra workloads of one class (host work: the GPU idles)."""
    ready = []
    for tag, kw, steps in extras_todo(args, world):
        if (kw['name'] in BIG) != big:
            continue
        kw = dict(kw)
        kw.pop('share', None)
        try:
            ready.append((tag, prepare(kw.pop('name'), args, rank, world,
                                       dist, **kw), steps))
        except Exception as exc:  # noqa: BLE001 - report, keep the line
            extra[tag] = {'error': f'{type(exc).__name__}: {exc}'}
    return ready


def measure_big_extras(args, rank, world, dist, extra):
    """The tens-of-GB workloads: prepare, measure, free -- one at a time (a
    row with `share` reuses the workload prepared just before it)."""
    import torch
    todo = [t for t in extras_todo(args, world) if t[1]['name'] in BIG]
    w, w_tag = None, None
    for n, (tag, kw, steps) in enumerate(todo):
        kw = dict(kw)
        t0 = time.perf_counter()
        try:
            share = kw.pop('share', None)
            if share is not None and share == w_tag:
                v = same_with(w, flags=kw.get('flags'), mode=kw.get('mode'),
                              mask=kw.get('mask', 'cells'),
                              tnl=kw.get('tnl', 0))
            else:
                if w is not None:
                    w.launch = w.fields = w.outs = w.full_field = None
                    w.remap = w.plan = w.full = None
                w = None          # (freed before the next one is built)
                gc.collect()
                torch.cuda.empty_cache()
                kw.pop('mask', None)
                kw.pop('tnl', None)
                v = w = prepare(kw.pop('name'), args, rank, world, dist, **kw)
                w_tag = tag
            measure_extras([(tag, v, steps)], args, dist, extra,
                           long_last=False)
            v.launch = None                   # (launch closes over v)
            v.full_field = None
            del v
        except Exception as exc:  # noqa: BLE001
            extra[tag] = {'error': f'{type(exc).__name__}: {exc}'}
        if isinstance(extra.get(tag), dict):
            extra[tag]['prepare_and_measure_s'] = time.perf_counter() - t0
    if w is not None:
        w.launch = w.fields = w.outs = w.full_field = None
        w.remap = w.plan = w.full = None
    del w
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
            extra[tag]['flags'] = w.flags
            extra[tag]['dtype'] = w.dtype
            extra[tag]['times'] = getattr(w, 'times', 8)
            extra[tag]['frac_of_peak'] = r['achieved_GBps'] / HBM_PEAK_GBPS
            extra[tag]['read_frac_of_peak'] = r['bytes_alg_read'] / (
                r['kernel_ms_mean'] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            # (the committed counter traffic is that of the (n_a, K) layout)
            traffic, _ = load_traffic(
                w.name, w.K, w.mode, w.locality,
                (r.get('schedule') or {}).get('family')) \
                if w.layout == 'nk' and w.dtype == 'f64' else (None, None)
            extra[tag]['traffic'] = traffic
            if dist is not None and w.sharded:
                # the slowest rank's kernel phase prices the WHOLE mapping's
                # algorithmic bytes against N x 8 TB/s
                tk = torch.tensor([r['kernel_ms_mean']], device='cuda',
                                  dtype=torch.float64)
                dist.all_reduce(tk, op=dist.ReduceOp.MAX)
                phase = float(tk.item())
                whole = w.full.algorithmic_bytes(
                    w.K, 4 if w.dtype == 'f32' else 8, w.emode)
                extra[tag]['multi'] = dict(
                    kernel_phase_ms=phase, bytes_alg_whole_mapping=whole,
                    frac_of_all_gpus=whole / (phase * 1e-3) / 1e9 /
                    (HBM_PEAK_GBPS * dist.get_world_size()),
                    packed_fraction=w.remap.packed_fraction(),
                    broadcast_ms=(w.exchange or {}).get('broadcast_ms'),
                    local_gather_ms=(w.exchange or {}).get(
                        'local_gather_ms'),
                    rows_this_rank=r['rows_this_rank'],
                    nnz_this_rank=r['nnz_this_rank'])
        except Exception as exc:  # noqa: BLE001
            extra[tag] = {'error': f'{type(exc).__name__}: {exc}'}


def multi_rows(extra):
    """`multi_gpu.workloads`: `tag: [kernel-phase ms of the slowest rank,
    fraction of N x 8 TB/s on the whole mapping's algorithmic bytes, packed
    fraction of the source rows this rank holds]`."""
    rows = {}
    for tag in MULTI_ROWS:
        e = extra.get(tag)
        if not isinstance(e, dict):
            continue
        mm = e.get('multi')
        if mm:
            rows[tag] = [round(mm['kernel_phase_ms'], 5),
                         round(mm['frac_of_all_gpus'], 4),
                         round(mm['packed_fraction'], 4)]
        elif 'error' in e:
            rows[tag] = 'error: ' + e['error'][:60]
    return rows


def replay_short_extras(ready, extra):
    """Launches the host cannot issue as fast as the GPU finishes them
    (< 50 us), replayed from a hipGraph -- after the metric is in hand."""
    for tag, w, _ in ready:
        e = extra.get(tag)
        if isinstance(e, dict) and e.get('kernel_ms_mean', 1.0) < 0.05:
            e['kernel_ms_graph_replay'] = graph_replay_ms(w.launch)


def kernel_of(e):
    """The kernel a workload's launches ran (for the side file)."""
    sched = e.get('schedule') or {}
    family = KERNEL_OF_FAMILY.get(sched.get('family'), 'spmm_*')
    rich = sched.get('rows_per_group') == 8
    if family == 'spmm_rowgroup' and e.get('mode') == 'masked' and \
            (e.get('flags') or 0) & 16 and rich and e['K'] > 128:
        family = ('spmm_cellshare' if sched.get('shared_by') and
                  e.get('dtype', 'f64') == 'f64' else
                  'spmm_groupmask') + ' (REMAP_FLAG_CELL_MASKS)'
    elif family == 'spmm_rowgroup' and e.get('mode') == 'masked' and \
            (e.get('flags') or 0) & 32 and rich and e['layout'] == 'tnl':
        family = ('spmm_timeshare' if sched.get('shared_by') else
                  'spmm_grouptime') + ' (REMAP_FLAG_BATCH_MASKS)'
    elif family == 'spmm_rowgroup' and rich and sched.get('shared_by') and \
            e.get('mode') != 'masked' and \
            (e['K'] >= 104 or 34 <= e['K'] <= 64) and \
            e.get('dtype', 'f64') == 'f64':
        family = 'spmm_groupshare' if e['K'] >= 104 else 'spmm_narrowshare'

    if sched.get('long_rows'):
        return family + ' + spmm_patchcell (long rows apart)'
    if e['layout'] == 'tn':
        return 'spmm_patchcell'
    if e['layout'] == 'tnl' and 4 <= e['K'] // e.get('times', 8) < 16:
        return 'spmm_patch'
    if e['K'] <= 32:
        return 'spmm_rowlane'
    return family


def workload_rows(extra):
    """
    `roofline.workloads`: `tag: [ms per launch, fraction of 8 TB/s, shader
    clock in MHz behind the timed launches]` per other workload (launches
    shorter than the host's ~12 us per call: as replayed from a hipGraph when
    that was measured).  Everything else about a
    workload is in the side file.
    """
    rows = {}
    for tag, e in extra.items():
        if not isinstance(e, dict):
            continue
        if 'kernel_ms_mean' not in e:
            if 'error' in e:
                rows[tag] = 'error: ' + e['error'][:60]
            continue
        ms = e.get('kernel_ms_graph_replay') or e['kernel_ms_mean']
        rows[tag] = [round(ms, 5), round(
            e['bytes_alg'] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)]
        if e.get('clock_mhz'):
            rows[tag].append(int(round(e['clock_mhz'])))
    return rows


def details_of(extra):
    """The side file's per-workload records: every measured field plus the
    derived ones the line used to carry."""
    out = {}
    for tag, e in extra.items():
        if not isinstance(e, dict) or 'kernel_ms_mean' not in e:
            out[tag] = e
            continue
        d = dict(e)
        d['kernel'] = kernel_of(e)
        if e.get('kernel_ms_graph_replay'):
            d['frac_graph_replay'] = e['bytes_alg'] / (
                e['kernel_ms_graph_replay'] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        if e.get('traffic'):
            d['traffic_ratio'] = e['traffic'] / e['bytes_alg']
        out[tag] = d
    return out


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


#: ... and the sharded tens-of-GB workloads of an N > 1 run
BIG_TIMEOUT_S = int(os.environ.get('BENCH_BIG_TIMEOUT_S', 420))


#: the line's size limit (the driver keeps the last 8 KB of stdout)
LINE_LIMIT = 4096


def rnd(x, digits=6):
    """Numbers in the line: `digits` significant digits are plenty."""
    if isinstance(x, float):
        return float(f'{x:.{digits}g}')
    return x


def compose_line(args, res, world, ceiling, cpu, extra, pipelined,
                 status='ok', details_path=None):
    """
    (line, details): the one JSON line -- metric, value, config, roofline
    (with `workloads: {tag: [ms, frac]}`), cpu_baseline, multi_gpu, status:
    under LINE_LIMIT bytes whatever was measured -- and the side file's
    content (everything else).
    """
    K = res['K']
    traffic, traffic_src = load_traffic(
        args.workload, K, res['mode'], args.locality,
        (res.get('schedule') or {}).get('family'))
    kernel_ms = res.get('kernel_ms_mean_max_rank', res['kernel_ms_mean'])
    achieved = res['bytes_alg'] / (kernel_ms * 1e-3) / 1e9
    multi = None
    if res['exchange'] is not None:
        full_multi = dict(res['exchange'])
        full_multi.update(pipelined or {})
        full_multi['kernel_phase_ms'] = kernel_ms
        keep = ('ranks', 'backend', 'exchange', 'broadcast_ms', 'packed_ms',
                'kernel_phase_ms', 'packed_fraction_of_broadcast',
                'local_gather_ms', 'pipelined_alltoall_ms_per_K_fields',
                'pipelined_broadcast_ms_per_K_fields',
                'optional_measurements', 'packed_error')
        multi = {k: (rnd(v) if not isinstance(v, str) else v[:100])
                 for k, v in full_multi.items() if k in keep}
        if multi_rows(extra):
            multi['workloads'] = multi_rows(extra)
    else:
        full_multi = None
    family = KERNEL_OF_FAMILY.get(res['schedule'].get('family'), 'spmm_*')
    line = {
        'metric': 'dst cell-fields/s (dst cells x batched fields per second)'
                  ' + HBM GB/s, EC30to60 MPAS -> 0.5deg lat-lon, 512 batched '
                  'fp64 fields',
        'value': rnd(res['cell_fields_per_s'], 9),
        'unit': 'dst cell-fields/s',
        'n_gpus': world,
        'steps': res['steps'],
        'warmup': res['warmup'],
        'ms_per_step': rnd(res['ms_per_step'], 9),
        'higher_is_better': True,
        'scaling': 'strong',   # the problem is fixed; rows (or fields) are
        #                        divided over the ranks
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': f"{args.workload}: {res['title']}"[:100],
            'n_a': res['n_a'], 'n_b': res['n_b'], 'n_s': res['n_s_file'],
            'nnz_csr': res['nnz_csr'], 'fields_K': K,
            'mode': res['mode'], 'layout': res['layout'],
            'locality': args.locality,
            'sharding': 'none' if world == 1 else
            (f'dst rows over {world} GPUs, packed column space, X '
             f'distributed once (RCCL) before the timed region'
             if args.shard == 'rows' else
             f'fields over {world} GPUs, no collective'),
            'buffer_sets_rotated': args.sets,
            'bitwise_mode': not (args.flags & 1),
        },
        'dst_cells_per_s_per_512_batch': rnd(
            res['dst_cells_per_s_per_batch']),
        'roofline': {
            'bound': 'hbm',
            'achieved': rnd(achieved),
            'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s',
            'frac': rnd(achieved / HBM_PEAK_GBPS),
            'traffic': traffic,
            'traffic_source': traffic_src and traffic_src[:80],
            'kernel': family + ' (remap_apply_f64)',
            'kernel_ms_mean': rnd(kernel_ms),
            'clock_mhz': rnd(res.get('clock_mhz'), 4),
            'kernel_ms_median': rnd(res['kernel_ms_median']),
            'kernel_ms_steady_100_more': rnd(
                res.get('kernel_ms_steady_100_more')),
            'bytes_alg_per_launch': res['bytes_alg'],
            'read_frac_of_peak': rnd(res['bytes_alg_read'] / (
                kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS),
            'measured_copy_ceiling_GBps': rnd(ceiling),
            'workloads': workload_rows(extra),
        },
        'cpu_baseline': None if cpu is None else {
            k: (rnd(v) if not isinstance(v, str) else v[:240])
            for k, v in cpu.items()
            if k in ('value', 'unit', 'cores', 'kind', 'sample', 'seconds',
                     'scipy_value', 'port_value', 'host_cpus', 'host_model')},
        'multi_gpu': multi,
        'status': status,
        'details': details_path,
    }
    pcie = extra.get('pcie_inclusive') if isinstance(extra, dict) else None
    if isinstance(pcie, dict):
        # host numpy in -> host numpy out (never `value`): [ms, GB/s]
        line['pcie_inclusive'] = [rnd(pcie['ms']), rnd(pcie['GBps'])] \
            if 'ms' in pcie else 'error: ' + pcie.get('error', '')[:60]
    details = {
        'line': None,       # (filled in by write_details)
        'result': res,
        'schedule': res['schedule'],
        'touched_frac': res['touched_frac'],
        'plan_build_s': res['plan_build_s'],
        'traffic_note': 'FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 per '
        'launch (MI355X_MICROARCH.md): the L2\'s fabric-side bytes, '
        'Infinity Cache hits included -- an upper bound on HBM bytes',
        'traffic_source': traffic_src,
        'measurement_order': 'metric first' if args.metric_first else
        'everything prepared first; then copy ceiling, extras, metric '
        'workload timed back to back',
        'cpu_baseline': cpu,
        'multi_gpu': full_multi,
        'workloads': details_of(extra),
    }
    # whatever was measured, the line stays parseable by the driver: rows
    # are dropped (they stay in the side file) before the limit is crossed
    rows = line['roofline']['workloads']
    while len(json.dumps(line)) > LINE_LIMIT and rows:
        rows.pop(next(reversed(rows)))
        line['roofline']['workloads_truncated'] = True
    return line, details


def write_details(path, line, details):
    """The side file; a box where it cannot be written still gets its line."""
    if path is None:
        return
    details['line'] = line
    try:
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path, 'w') as f:
            json.dump(details, f, indent=1, default=str)
    except OSError as exc:
        print(f'bench.py: side file {path}: {exc}', file=sys.stderr)


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    import torch
    t_start = time.perf_counter()
    rank, world, local, dist = init_dist(args)
    device = torch.device('cuda', local)
    from pyremap_amd import engine
    engine.require_gpu()
    details_path = args.details or os.path.join(_REPO, 'gpurun_out',
                                                'bench_extra.json')
    shown_path = os.path.relpath(details_path, _REPO) \
        if details_path.startswith(_REPO) else details_path
    phases = {}

    def mark(name, since):
        torch.cuda.synchronize()
        phases[name] = round(time.perf_counter() - since, 3)
        return time.perf_counter()

    extra = {}
    # The metric workload and the small extras are PREPARED first (host
    # work, GPU mostly idle); the tens-of-GB workloads come and go one at a
    # time in between; then all remaining measurements run back to back, the
    # metric workload last.
    t = mark('init_s', t_start)
    main_w = prepare(args.workload, args, rank, world, dist)
    t = mark('prepare_metric_workload_s', t)
    ready = prepare_extras(args, rank, world, dist, extra, big=False)
    t = mark('prepare_small_extras_s', t)
    if world == 1:
        measure_big_extras(args, rank, world, dist, extra)
        t = mark('big_extras_s', t)
    # (N > 1: the sharded tens-of-GB workloads run AFTER the metric is in
    # hand, under the watchdog -- their broadcasts move 15-30 GB over a
    # fabric no test could exercise)
    ceiling = copy_ceiling(device)
    t = mark('copy_ceiling_s', t)
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
    t = mark('small_extras_and_metric_s', t)
    # per-rank kernel numbers -> the slowest rank prices the roofline
    if dist is not None:
        tk = torch.tensor([res['kernel_ms_mean']], device=device,
                          dtype=torch.float64)
        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        res['kernel_ms_mean_max_rank'] = float(tk.item())
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
                    'timed out: the packed / pipelined exchange or a '
                    'sharded extra workload did not return')
                res['exchange'] = late
                line, details = compose_line(
                    args, res, world, ceiling, None, extra, None,
                    status='exchange_hung', details_path=shown_path)
                write_details(details_path, line, details)
                print_line(line)
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
        t = mark('exchange_measurements_s', t)
        # the sharded headline / config 4 / config 5 (multi_gpu.workloads),
        # under a watchdog of their own
        if world > 1:
            def bail_big():
                # The metric and its exchange measurements are complete and
                # clean; what did not return is an EXTRA workload (15-30 GB
                # broadcasts, or a kernel that hangs).  The line is printed
                # all the same -- the scaling record survives, `status` and
                # the missing rows of multi_gpu.workloads say what happened
                # -- but a process that leaves a collective or a kernel
                # behind does not end like a clean run: exit code 4 (3 is
                # the hung exchange of the metric itself).
                if rank == 0:
                    late = dict(res['exchange'] or {})
                    late['optional_measurements'] = (
                        'a sharded extra workload (headline / config 4 / '
                        'config 5) did not return in time')
                    res['exchange'] = late
                    line, details = compose_line(
                        args, res, world, ceiling, None, extra, pipelined,
                        status='extras_timed_out', details_path=shown_path)
                    write_details(details_path, line, details)
                    print_line(line)
                os._exit(4)
            watchdog = threading.Timer(BIG_TIMEOUT_S, bail_big)
            watchdog.daemon = True
            watchdog.start()
            if os.environ.get('BENCH_TEST_HANG_BIG'):   # exercises it
                time.sleep(10 ** 6)
            measure_big_extras(args, rank, world, dist, extra)
            barrier(dist)
            watchdog.cancel()
            t = mark('big_extras_s', t)
    elif world > 1:
        measure_big_extras(args, rank, world, dist, extra)
        t = mark('big_extras_s', t)
    del ready

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(main_w.full, main_w.m, main_w.fields[0],
                           res['mode'], args.cpu_seconds)
        t = mark('cpu_baseline_s', t)
    if world == 1 and not args.no_extra and args.workload == 'config3' \
            and not main_w.sharded:
        try:
            extra['pcie_inclusive'] = pcie_row(main_w)
        except Exception as exc:  # noqa: BLE001
            extra['pcie_inclusive'] = {'error': f'{type(exc).__name__}: '
                                       f'{exc}'}
        t = mark('pcie_row_s', t)
    if world == 1 and args.all_workloads and not args.no_extra and \
            args.workload == 'config3':
        main_w.fields = main_w.outs = None
        torch.cuda.empty_cache()
        try:
            extra['host_buffers_pcie_inclusive'] = pcie_inclusive(args)
        except Exception as exc:  # noqa: BLE001
            extra['host_buffers_pcie_inclusive'] = {
                'error': f'{type(exc).__name__}: {exc}'}
        t = mark('pcie_inclusive_s', t)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    phases['total_s'] = round(time.perf_counter() - t_start, 3)
    line, details = compose_line(args, res, world, ceiling, cpu, extra,
                                 pipelined, details_path=shown_path)
    details['phases_s'] = phases
    write_details(details_path, line, details)
    # the JSON line is the LAST thing on stdout: every rank is done first
    # (RCCL prints a version banner on stdout when its communicator comes up
    # or goes down, whichever happens to be later)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    print_line(line)


if __name__ == '__main__':
    main()
