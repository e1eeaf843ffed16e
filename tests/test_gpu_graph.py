"""
Launches captured in a hipGraph and replayed on NEW data: `remap_apply_f64`
and `remap_scan_nan` launch on the caller's stream and neither allocate nor
synchronise, so a series of calls on fixed buffers -- the time loop of an
analysis, many small variables -- is captured once (torch.cuda.CUDAGraph)
and replayed, free of the ~12 us per call the host needs to issue one.  Every
kernel family the dispatcher picks on its own is in the captured series; the
replay is checked against the oracle on data that did not exist at capture
time.
"""
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


def _plans(dev):
    from oracle import oracle
    from pyremap_amd import engine, synthetic
    out = []
    for m, dims in (
            (synthetic.conservative_map(6000, (50, 80), 1, 7, seed=3,
                                        device=dev, locality='mesh'),
             (50, 80)),
            (synthetic.bilinear_map((30, 40), (90, 110), seed=4, device=dev),
             (90, 110))):
        plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                              m.n_a, m.n_b, device=dev)
        choice = plan.auto_schedule(dims)
        rowptr, col, val = plan.to_host_csr()
        csr = oracle.OracleCSR(rowptr, col, val, (m.n_b, m.n_a))
        out.append((m, dims, plan, csr, m.frac_b.cpu().numpy(), choice))
    return out


# (shape with -1 = the source cells, remap axis, mode)
CALLS = [((-1, 96), 0, 'fracb'),      # row groups / LDS patches
         ((-1, 96), 0, 'masked'),
         ((-1,), 0, 'fracb'),         # one 2-D field: lane per row
         ((1, -1), 1, 'masked'),
         ((40, -1), 1, 'fracb'),      # (Time, nCells): lanes across rows
         ((40, -1), 1, 'masked'),
         ((6, -1, 10), 1, 'fracb'),   # short level runs: small LDS patches
         ((3, -1, 61), 1, 'raw'),     # odd level count
         ((-1, 24), 0, 'auto'),       # NaN scan + two gated launches
         ((12, -1), 1, 'auto')]


def test_launches_replay_from_a_hip_graph(dev):
    from oracle import oracle
    from pyremap_amd import engine
    rng = np.random.default_rng(11)
    for m, dims, plan, csr, frac_b, choice in _plans(dev):
        emodes = {'fracb': engine.MODE_FRACB, 'masked': engine.MODE_MASKED,
                  'raw': engine.MODE_RAW}
        xs, ys, flags = [], [], []
        for shape, axis, mode in CALLS:
            shape = tuple(m.n_a if v < 0 else v for v in shape)
            dtype = torch.float32 if len(shape) == 3 else torch.float64
            xs.append(torch.zeros(shape, device=dev, dtype=dtype))
            flags.append(torch.zeros(1, dtype=torch.int32, device=dev))

        def series(outs):
            got = []
            for n, (x, (shape, axis, mode)) in enumerate(zip(xs, CALLS)):
                if mode == 'auto':
                    flags[n].zero_()
                    got.append(engine.remap_tensor_auto_mode(
                        plan, dims, x, [axis], 0.01, flag=flags[n],
                        out=outs[n] if outs else None))
                else:
                    got.append(engine.remap_tensor(
                        plan, dims, x, [axis], emodes[mode], threshold=0.01,
                        out=outs[n] if outs else None))
            return got
        # an eager pass first: output buffers, and the patch plans the short
        # layouts build on first use (allocations and one readback each)
        ys = series(None)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            series(ys)
        for round_ in range(2):
            # new data in the captured buffers; the second round flips which
            # 'auto' call holds NaNs (the gate is part of the replay)
            host = []
            for n, (x, (shape, axis, mode)) in enumerate(zip(xs, CALLS)):
                h = rng.standard_normal(tuple(x.shape)).astype(
                    np.float32 if x.dtype == torch.float32 else np.float64)
                holes = mode == 'masked' or \
                    (mode == 'auto' and (n + round_) % 2 == 0)
                if holes:
                    dead = rng.random(m.n_a) < 0.25
                    h[(slice(None),) * axis + (dead,)] = np.nan
                host.append((h, holes))
                x.copy_(torch.from_numpy(h))
            for y in ys:
                y.fill_(-7.0)
            graph.replay()
            torch.cuda.synchronize()
            for n, ((h, holes), (shape, axis, mode)) in enumerate(
                    zip(host, CALLS)):
                if mode == 'raw':
                    # the bare product A . X, in the output's axis order
                    X = np.moveaxis(h, axis, 0).reshape(m.n_a, -1)
                    flat = oracle.csr_matvecs(csr, X.astype(np.float64))
                    lead, tail = h.shape[:axis], h.shape[axis + 1:]
                    want = np.moveaxis(
                        flat.reshape(tuple(dims) + lead + tail),
                        list(range(len(dims))),
                        list(range(axis, axis + len(dims))))
                else:
                    masked = mode == 'masked' or (mode == 'auto' and holes)
                    arg = np.ma.masked_array(h, np.isnan(h)) if masked else h
                    want = np.ma.filled(oracle.remap_numpy_array(
                        csr, frac_b, dims, arg, [axis],
                        0.01 if masked else None), np.nan)
                assert_bitwise(ys[n].cpu().numpy(), want,
                               f'{choice["family"]} call {n} {shape} {mode} '
                               f'round {round_}')


def test_plan_handle_launches_are_capturable(dev):
    """The C plan handle (`remap_plan_apply`) replayed from a graph."""
    import ctypes

    from oracle import oracle
    from pyremap_amd import engine, synthetic
    lib = engine.load_library()
    m = synthetic.conservative_map(5000, (40, 60), 1, 6, seed=8, device=dev,
                                   locality='mesh')
    mm = m.numpy()
    csr = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b,
                            m.n_a)
    dims = (ctypes.c_int64 * 2)(40, 60)
    row = torch.as_tensor(mm['row'], dtype=torch.int32, device=dev)
    col = torch.as_tensor(mm['col'], dtype=torch.int32, device=dev)
    S = torch.as_tensor(mm['S'], dtype=torch.float64, device=dev)
    fb = torch.as_tensor(mm['frac_b'], dtype=torch.float64, device=dev)
    handle = ctypes.c_void_p()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.remap_plan_create(
        m.n_b, m.n_a, S.numel(), row.data_ptr(), col.data_ptr(),
        S.data_ptr(), 1, fb.data_ptr(), 0, dims, 2, stream,
        ctypes.byref(handle))
    assert rc == 0, lib.remap_last_error()
    try:
        assert lib.remap_plan_prepare_short_runs(handle, stream) == 0
        torch.cuda.synchronize()
        x = torch.zeros((30, m.n_a), device=dev, dtype=torch.float64)
        y = torch.empty((30, m.n_b), device=dev, dtype=torch.float64)
        f = engine._Field()
        f.X, f.x_dtype = x.data_ptr(), engine.DTYPE_F64
        f.n_batch, f.k_inner = 30, 1
        f.x_row_stride, f.x_batch_stride = 1, m.n_a
        f.Y = y.data_ptr()
        f.y_row_stride, f.y_batch_stride = 1, m.n_b
        f.mode, f.threshold = engine.MODE_FRACB, 0.0
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            assert lib.remap_plan_apply(handle, ctypes.byref(f), s) == 0
        h = np.random.default_rng(2).standard_normal((30, m.n_a))
        x.copy_(torch.from_numpy(h))
        graph.replay()
        torch.cuda.synchronize()
        want = np.ma.filled(oracle.remap_numpy_array(
            csr, mm['frac_b'], (40, 60), h, [1], None), np.nan)
        assert_bitwise(y.cpu().numpy().reshape(30, 40, 60), want,
                       'plan handle replayed')
        # `remap_plan_apply_auto` (memset of the flags, the scan, the gated
        # launches) is capturable too: one capture, replayed on a field
        # without NaNs and on one with (the branch is taken on the device)
        kinds = torch.zeros(4, dtype=torch.int32, device=dev)
        f.threshold = 0.2
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph2):
            s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            assert lib.remap_plan_apply_auto(
                handle, ctypes.byref(f), x.numel(), kinds.data_ptr(),
                s) == 0, lib.remap_last_error()
        for with_nan in (False, True, False):
            h = np.random.default_rng(3).standard_normal((30, m.n_a))
            if with_nan:
                h[:, ::7] = np.nan
            x.copy_(torch.from_numpy(h))
            graph2.replay()
            torch.cuda.synchronize()
            arg = np.ma.masked_array(h, np.isnan(h)) if with_nan else h
            want = np.ma.filled(oracle.remap_numpy_array(
                csr, mm['frac_b'], (40, 60), arg, [1], 0.2), np.nan)
            assert int(kinds[0]) == int(with_nan), kinds.tolist()
            assert_bitwise(y.cpu().numpy().reshape(30, 40, 60), want,
                           f'apply_auto replayed, NaNs: {with_nan} '
                           f'{kinds.tolist()}')
    finally:
        lib.remap_plan_destroy(handle)
