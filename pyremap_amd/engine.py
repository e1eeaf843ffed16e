"""
Host side of the MI355X weight-application engine: the ctypes binding of
``libremap_hip.so`` (``include/remap_hip.h``) plus the device-resident plan.

This is the layer that replaces, for ``Remapper.remap_numpy``:

* ``remap_numpy.py:134-137`` -- ``RemapPlan.from_triplets`` (COO -> CSR on
  the device, cached on the Remapper where the reference caches ``_matrix``);
* ``remap_numpy.py:223-297`` -- ``remap_array`` (``_remap_numpy_array``):
  the permute/flatten and unflatten/unpermute steps become strides handed to
  the kernel, the SpMM + normalisation + masking is one fused HIP launch.

PyTorch is used for device memory, streams and (in ``parallel.py``)
``torch.distributed``; the arithmetic is in the HIP library.  There is no CPU
fallback: without the library or without a GPU every compute entry point
raises.
"""
import ctypes
import os

import numpy as np

from pyremap_amd import _build

MODE_RAW = 0
MODE_FRACB = 1
MODE_MASKED = 2

FLAG_FMA = 1
FLAG_TUNE_HINT = 4
FLAG_TREE = 8
FLAG_CELL_MASKS = 16   # masked mode, a hint: validity is per source cell
FLAG_BATCH_MASKS = 32  # masked mode, a hint: the same mask in every batch

#: long rows apart (RemapPlan._split_long_rows): up to this many fields the
#: long rows run one wave per (row, few columns) -- family 9 -- beyond it on
#: the LDS-staged lanes-across-rows kernel (family 7)
LONG_WAVE_FIELDS = 16
#: long rows per workgroup (one wave each) of family 11 (spmm_longwave), which
#: takes the long rows' launch from LONG_WAVE_FIELDS + 1 to LONG_WAVE_MAX
#: fields; 0: family 7 there too.  1 deg -> 0.5 deg with pole caps, the long
#: rows' launch replayed, us, family 7 / family 11 with 4 / 6 / 8 rows:
#: K = 24 26.5 / 23.9 / 20.0 / 19.2, 64 26.2 / 23.4 / 19.8 / 19.1, 128 38.2 /
#: 28.3 / 27.4 / 36.3, 256 50.7 / 51.4 / 50.0 / 54.2, 512 61.3 / 98 / 95 / 107
LONG_WAVE_ROWS = 6
LONG_WAVE_MAX = 128
#: fields whose contiguous run behind the source axes is shorter than this
#: (and that come in several batches) take the lanes-across-rows kernels
CELL_MAX_RUN = 4

DTYPE_F64 = 0
DTYPE_F32 = 1

ABI_VERSION = 25

#: readable pad entries kept behind col/val (remap_csr.csr_pad)
CSR_PAD = 8
#: LDS a workgroup of the lanes-across-rows kernel may take (kPatchLdsMax)
CELL_LDS_MAX = 160 * 1024
_LONG_TT = None   # (experiments: fields per lane of the long-row launch)
_LONG_WAVE_TT = None   # (tests: columns per wave of family 9, 0 = auto)
_CELL_TUNE = None      # (tools/tn_probe.py: [TT, one-chunk kernel?] of family 7)
_RUNS_TUNE = None      # (tools/runs_probe.py: the batch-at-a-time launch's tune)

#: every symbol ``include/remap_hip.h`` declares
EXPORTS = (
    'remap_abi_version', 'remap_arch', 'remap_last_error',
    'remap_device_count', 'remap_apply_f64', 'remap_csr_from_coo_workspace',
    'remap_csr_from_coo', 'remap_stream_copy', 'remap_scan_nan',
    'remap_scan_nan_kinds', 'remap_scan_nan_layout',
    'remap_groups_workspace', 'remap_groups_build', 'remap_share_build',
    'remap_patches_workspace', 'remap_patches_build',
    'remap_schedule_sizes', 'remap_schedule_auto',
    'remap_plan_create', 'remap_plan_destroy', 'remap_plan_query',
    'remap_plan_apply', 'remap_plan_apply_auto',
    'remap_pack_columns_workspace', 'remap_pack_columns',
    'remap_gather_rows', 'remap_plan_prepare_short_runs',
    'remap_clock_probe',
)


class _CSR(ctypes.Structure):
    _fields_ = [
        ('n_rows', ctypes.c_int64),
        ('n_cols', ctypes.c_int64),
        ('nnz', ctypes.c_int64),
        ('rowptr', ctypes.c_void_p),
        ('col', ctypes.c_void_p),
        ('val', ctypes.c_void_p),
        ('max_row_nnz', ctypes.c_int64),
        ('csr_pad', ctypes.c_int64),
    ]


class _ApplyArgs(ctypes.Structure):
    _fields_ = [
        ('A', _CSR),
        ('row_begin', ctypes.c_int64),
        ('row_end', ctypes.c_int64),
        ('X', ctypes.c_void_p),
        ('x_dtype', ctypes.c_int32),
        ('mode', ctypes.c_int32),
        ('x_row_stride', ctypes.c_int64),
        ('x_batch_stride', ctypes.c_int64),
        ('Y', ctypes.c_void_p),
        ('y_row_stride', ctypes.c_int64),
        ('y_batch_stride', ctypes.c_int64),
        ('n_batch', ctypes.c_int64),
        ('k_inner', ctypes.c_int64),
        ('frac_b', ctypes.c_void_p),
        ('threshold', ctypes.c_double),
        ('mask_out', ctypes.c_void_p),
        ('row_order', ctypes.c_void_p),
        ('patch_ptr', ctypes.c_void_p),
        ('patch_ucol', ctypes.c_void_p),
        ('patch_rowptr', ctypes.c_void_p),
        ('patch_lidx', ctypes.c_void_p),
        ('patch_val', ctypes.c_void_p),
        ('patch_rows', ctypes.c_int32),
        ('patch_umax', ctypes.c_int32),
        ('patch_emax', ctypes.c_int32),
        ('patch_row_bytes', ctypes.c_int32),
        ('n_patches', ctypes.c_int64),
        ('group_meta', ctypes.c_void_p),
        ('group_col', ctypes.c_void_p),
        ('group_w', ctypes.c_void_p),
        ('group_mask', ctypes.c_void_p),
        ('group_rid', ctypes.c_void_p),
        ('group_frac', ctypes.c_void_p),
        ('n_groups', ctypes.c_int64),
        ('group_rows', ctypes.c_int32),
        ('group_reserved', ctypes.c_int32),
        ('gate', ctypes.c_void_p),
        ('gate_value', ctypes.c_int32),
        ('flags', ctypes.c_uint32),
        ('tune', ctypes.c_int32 * 8),
        ('x_src_fold', ctypes.c_int64),
        ('x_outer_stride', ctypes.c_int64),
        ('patch_ell_base', ctypes.c_void_p),
        ('strips', ctypes.c_void_p),
        ('share_meta', ctypes.c_void_p),
        ('share_col', ctypes.c_void_p),
        ('share_mask', ctypes.c_void_p),
        ('share_waves', ctypes.c_int32),
        ('share_reserved', ctypes.c_int32),
    ]


class _Strips(ctypes.Structure):         # struct remap_strips
    _fields_ = [
        ('n_units', ctypes.c_int64),
        ('steps_per_unit', ctypes.c_int32),
        ('rows_per_wave', ctypes.c_int32),
        ('ring_slots', ctypes.c_int32),
        ('depth', ctypes.c_int32),
        ('meta_slot_bytes', ctypes.c_int32),
        ('waves', ctypes.c_int32),
        ('unit_steps', ctypes.c_void_p),
        ('arr_ptr', ctypes.c_void_p),
        ('arr_src', ctypes.c_void_p),
        ('arr_slot', ctypes.c_void_p),
        ('meta_ptr', ctypes.c_void_p),
        ('meta', ctypes.c_void_p),
    ]


class _Schedule(ctypes.Structure):
    _fields_ = [
        ('family', ctypes.c_int32),
        ('entry_rich', ctypes.c_int32),
        ('row_order', ctypes.c_void_p),
        ('patch_ptr', ctypes.c_void_p),
        ('patch_ucol', ctypes.c_void_p),
        ('patch_rowptr', ctypes.c_void_p),
        ('patch_lidx', ctypes.c_void_p),
        ('patch_val', ctypes.c_void_p),
        ('patch_rows', ctypes.c_int32),
        ('patch_umax', ctypes.c_int32),
        ('patch_emax', ctypes.c_int32),
        ('patch_row_bytes', ctypes.c_int32),
        ('n_patches', ctypes.c_int64),
        ('group_meta', ctypes.c_void_p),
        ('group_col', ctypes.c_void_p),
        ('group_w', ctypes.c_void_p),
        ('group_mask', ctypes.c_void_p),
        ('group_rid', ctypes.c_void_p),
        ('group_frac', ctypes.c_void_p),
        ('n_groups', ctypes.c_int64),
        ('group_rows', ctypes.c_int32),
        ('super_tile', ctypes.c_int32),
        ('tile_y', ctypes.c_int32),
        ('tile_x', ctypes.c_int32),
        ('ratio', ctypes.c_double),
        ('n_distinct', ctypes.c_int64),
        ('tune', (ctypes.c_int32 * 8) * 3),
        ('arena_used', ctypes.c_size_t),
        ('share_meta', ctypes.c_void_p),
        ('share_col', ctypes.c_void_p),
        ('share_mask', ctypes.c_void_p),
        ('share_waves', ctypes.c_int32),
        ('share_reserved', ctypes.c_int32),
        ('n_share_union', ctypes.c_int64),
    ]


class _PlanInfo(ctypes.Structure):     # struct remap_plan_info
    _fields_ = [
        ('n_a', ctypes.c_int64),
        ('n_b', ctypes.c_int64),
        ('nnz', ctypes.c_int64),
        ('max_row_nnz', ctypes.c_int64),
        ('family', ctypes.c_int32),
        ('group_rows', ctypes.c_int32),
        ('ratio', ctypes.c_double),
        ('device_bytes', ctypes.c_size_t),
        ('cell_patch_rows', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
    ]


class _Field(ctypes.Structure):        # struct remap_field
    _fields_ = [
        ('X', ctypes.c_void_p),
        ('x_dtype', ctypes.c_int32),
        ('mode', ctypes.c_int32),
        ('n_batch', ctypes.c_int64),
        ('k_inner', ctypes.c_int64),
        ('x_row_stride', ctypes.c_int64),
        ('x_batch_stride', ctypes.c_int64),
        ('Y', ctypes.c_void_p),
        ('y_row_stride', ctypes.c_int64),
        ('y_batch_stride', ctypes.c_int64),
        ('threshold', ctypes.c_double),
        ('mask_out', ctypes.c_void_p),
        ('gate', ctypes.c_void_p),
        ('gate_value', ctypes.c_int32),
        ('flags', ctypes.c_uint32),
    ]


_lib = None


class EngineError(RuntimeError):
    """A failure reported by libremap_hip.so (message from the C side)."""


def library_path():
    return _build.LIB_PATH


def load_library():
    """
    Load ``libremap_hip.so``; build it first if it is missing and hipcc is
    available.  Raises if the library cannot be provided -- there is no other
    implementation to fall back to.
    """
    global _lib
    if _lib is not None:
        return _lib
    # torch brings its own HIP runtime (torch/lib/libamdhip64.so); the
    # library's RUNPATH names the system one.  Whichever is loaded first
    # serves the whole process, and a process with BOTH initialised sees "no
    # ROCm-capable device" from the second: torch first, always.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = _build.LIB_PATH
    if not os.path.exists(path):
        if _build.find_hipcc() is None:
            raise EngineError(
                f'{path} is missing and hipcc is not available to build it; '
                f'run `python -c "import __graft_entry__ as g; g.build()"` '
                f'on a machine with ROCm')
        _build.build_library()
    lib = ctypes.CDLL(path)
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise EngineError(f'{path} lacks symbols {missing}')
    lib.remap_abi_version.restype = ctypes.c_int
    lib.remap_arch.restype = ctypes.c_char_p
    lib.remap_last_error.restype = ctypes.c_char_p
    lib.remap_device_count.restype = ctypes.c_int
    lib.remap_apply_f64.restype = ctypes.c_int
    lib.remap_apply_f64.argtypes = [ctypes.POINTER(_ApplyArgs),
                                    ctypes.c_void_p]
    lib.remap_csr_from_coo_workspace.restype = ctypes.c_int
    lib.remap_csr_from_coo_workspace.argtypes = [
        ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t)]
    lib.remap_csr_from_coo.restype = ctypes.c_int
    lib.remap_csr_from_coo.argtypes = [
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.remap_groups_workspace.restype = ctypes.c_int
    lib.remap_groups_workspace.argtypes = [
        ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t)]
    lib.remap_groups_build.restype = ctypes.c_int
    lib.remap_groups_build.argtypes = [
        ctypes.POINTER(_CSR), ctypes.c_void_p, ctypes.c_int32,
        ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.c_int32,
        ctypes.c_int32] + \
        [ctypes.c_void_p] * 9 + [ctypes.c_size_t, ctypes.c_void_p]
    lib.remap_share_build.restype = ctypes.c_int
    lib.remap_share_build.argtypes = [
        ctypes.POINTER(_CSR), ctypes.c_void_p, ctypes.c_int32] + \
        [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p]
    lib.remap_patches_workspace.restype = ctypes.c_int
    lib.remap_patches_workspace.argtypes = [
        ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t)]
    lib.remap_patches_build.restype = ctypes.c_int
    lib.remap_patches_build.argtypes = [
        ctypes.POINTER(_CSR), ctypes.POINTER(ctypes.c_int64), ctypes.c_int64,
        ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 8 + \
        [ctypes.c_size_t, ctypes.c_void_p]
    lib.remap_schedule_sizes.restype = ctypes.c_int
    lib.remap_schedule_sizes.argtypes = [
        ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t),
        ctypes.POINTER(ctypes.c_size_t)]
    lib.remap_schedule_auto.restype = ctypes.c_int
    lib.remap_schedule_auto.argtypes = [
        ctypes.POINTER(_CSR), ctypes.c_void_p,
        ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, ctypes.c_int64,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
        ctypes.POINTER(_Schedule), ctypes.c_void_p]
    lib.remap_plan_create.restype = ctypes.c_int
    lib.remap_plan_create.argtypes = [
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32,
        ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.remap_plan_destroy.restype = None
    lib.remap_plan_destroy.argtypes = [ctypes.c_void_p]
    lib.remap_plan_query.restype = ctypes.c_int
    lib.remap_plan_query.argtypes = [ctypes.c_void_p,
                                     ctypes.POINTER(_PlanInfo)]
    lib.remap_plan_prepare_short_runs.restype = ctypes.c_int
    lib.remap_plan_prepare_short_runs.argtypes = [ctypes.c_void_p,
                                                  ctypes.c_void_p]
    lib.remap_plan_apply.restype = ctypes.c_int
    lib.remap_plan_apply.argtypes = [ctypes.c_void_p,
                                     ctypes.POINTER(_Field), ctypes.c_void_p]
    lib.remap_plan_apply_auto.restype = ctypes.c_int
    lib.remap_plan_apply_auto.argtypes = [
        ctypes.c_void_p, ctypes.POINTER(_Field), ctypes.c_int64,
        ctypes.c_void_p, ctypes.c_void_p]
    lib.remap_pack_columns_workspace.restype = ctypes.c_int
    lib.remap_pack_columns_workspace.argtypes = [
        ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t)]
    lib.remap_pack_columns.restype = ctypes.c_int
    lib.remap_pack_columns.argtypes = [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_size_t, ctypes.c_void_p]
    lib.remap_gather_rows.restype = ctypes.c_int
    lib.remap_gather_rows.argtypes = [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p]
    lib.remap_scan_nan.restype = ctypes.c_int
    lib.remap_scan_nan.argtypes = [ctypes.c_void_p, ctypes.c_int32,
                                   ctypes.c_int64, ctypes.c_void_p,
                                   ctypes.c_void_p]
    lib.remap_scan_nan_kinds.restype = ctypes.c_int
    lib.remap_scan_nan_kinds.argtypes = [ctypes.c_void_p, ctypes.c_int32,
                                   ctypes.c_int64, ctypes.c_void_p,
                                   ctypes.c_void_p]
    lib.remap_scan_nan_layout.restype = ctypes.c_int
    lib.remap_scan_nan_layout.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64,
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p]
    lib.remap_stream_copy.restype = ctypes.c_int
    lib.remap_stream_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_size_t, ctypes.c_void_p]
    lib.remap_clock_probe.restype = ctypes.c_int
    lib.remap_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int32,
                                      ctypes.c_void_p]
    if lib.remap_abi_version() != ABI_VERSION:
        raise EngineError(
            f'{path} has ABI {lib.remap_abi_version()}, expected '
            f'{ABI_VERSION}; rebuild it')
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().remap_last_error().decode('utf-8', 'replace')
        raise EngineError(f'{what} failed ({rc}): {msg}')


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise EngineError(
            'no HIP device is visible: the remapping engine runs on MI355X '
            'only (there is no CPU implementation in pyremap_amd)')
    load_library()
    return torch


def _stream_ptr(device):
    torch = _torch()
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ---------------------------------------------------------------------------
# the plan: device-resident CSR + frac_b
# ---------------------------------------------------------------------------

class RemapPlan:
    """
    Device-resident weights of one mapping file (or of a row shard of it):
    what the reference keeps as ``remapper._matrix`` plus ``frac_b``.

    Attributes
    ----------
    n_a, n_b : int
        source size and number of destination rows HELD by this plan
    n_b_global : int
        destination size of the whole mapping
    row_offset : int
        global index of this plan's first row (0 unless sharded)
    rowptr, col, val, frac_b : torch.Tensor
        int64[n_b + 1], int32[nnz], float64[nnz], float64[n_b] on ``device``
    """

    def __init__(self, n_a, n_b, rowptr, col, val, frac_b, row_offset=0,
                 n_b_global=None):
        self.n_a = int(n_a)
        self.n_b = int(n_b)
        self.rowptr = rowptr
        self.frac_b = frac_b
        self.row_offset = int(row_offset)
        self.n_b_global = int(n_b if n_b_global is None else n_b_global)
        self.device = val.device
        torch = _torch()
        self.nnz = int(rowptr[-1]) if rowptr.numel() else 0
        #: col/val carry CSR_PAD readable entries behind the last one so a
        #: row's entries can be fetched 8 at a time (scalar-cache kernels)
        pad = int(val.shape[0]) - self.nnz
        if pad < CSR_PAD:
            col = torch.cat([col[:self.nnz], torch.zeros(
                CSR_PAD, dtype=col.dtype, device=col.device)])
            val = torch.cat([val[:self.nnz], torch.zeros(
                CSR_PAD, dtype=val.dtype, device=val.device)])
        self.csr_pad = int(val.shape[0]) - self.nnz
        self._col_padded = col
        self._val_padded = val
        self.col = col[:self.nnz]
        self.val = val[:self.nnz]
        #: entries of the longest row (kernel selection)
        self.max_row_nnz = int((rowptr[1:] - rowptr[:-1]).max()) \
            if self.n_b > 0 else 0
        self._touched = None
        self._arena = None
        self._row_last_col = None
        self._extent_cache = {}
        #: launch tuning used when a call passes none (set by auto_schedule)
        self.default_tune = None
        #: destination grid the schedules were built for (auto_schedule)
        self._grid_dims = None
        #: lazily built patch plan for the lanes-across-rows kernel that
        #: serves (Time, nCells)-like layouts (see cell_patches)
        self._cell = None
        #: lazily built small LDS patches for fields with short level runs
        #: (see run_patches)
        self._runs = None
        self._run_cells = None
        self._wave = None
        # schedule attributes are properties: every assignment invalidates
        # the prefilled argument block launches start from (_prefilled)
        self._sched_version = 0
        self._args_cache = {}
        #: optional LDS-staging schedule (see build_patches)
        self.patches = None
        #: optional row-group schedule (see build_groups)
        self.groups = None
        #: optional int32 permutation of the rows: the order in which work
        #: slots visit them (scheduling only; see set_row_order)
        self.row_order = None
        #: (short, long) sibling plans when a few rows hold a large share of
        #: the entries (see _split_long_rows); None otherwise
        self._split = None
        #: optional strip schedule of kernel family 8 (see build_strips)
        self.strips = None

    def _sched_property(name):   # noqa: N805 - class-body helper
        def get(self):
            return self.__dict__.get('_' + name)

        def put(self, value):
            self.__dict__['_' + name] = value
            self._sched_version += 1
        return property(get, put)

    patches = _sched_property('patches')
    groups = _sched_property('groups')
    strips = _sched_property('strips')
    row_order = _sched_property('row_order')
    del _sched_property

    def _prefilled(self, whole, cell=False):
        """
        A fresh ``remap_apply_args`` with everything that belongs to the plan
        filled in -- the CSR and, for a launch over the whole row range, the
        row order and the schedule that goes with it (``cell``: the patch
        plan of the lanes-across-rows kernel instead) -- copied from a block
        built once per schedule version (a Dataset of small variables is
        launch-overhead-bound: 2 launches per variable).
        """
        key = (whole, cell)
        hit = self._args_cache.get(key)
        if hit is not None and hit[0] == self._sched_version:
            return _ApplyArgs.from_buffer_copy(hit[1])
        args = _ApplyArgs()
        args.A.n_rows = self.n_b
        args.A.n_cols = self.n_a
        args.A.nnz = self.nnz
        args.A.rowptr = self.rowptr.data_ptr()
        args.A.col = self.col.data_ptr()
        args.A.val = self.val.data_ptr()
        args.A.max_row_nnz = self.max_row_nnz
        args.A.csr_pad = self.csr_pad
        order = self.row_order
        if whole and cell:
            q = self._runs if cell == 'runs' else \
                self._run_cells if cell == 'run_cells' else \
                self._wave if cell == 'wave' else self._cell
            args.row_order = q['order'].data_ptr() \
                if q['order'] is not None else None
            args.patch_ptr = q['ptr'].data_ptr()
            args.patch_ucol = q['ucol'].data_ptr()
            args.patch_rowptr = q['rowptr'].data_ptr()
            args.patch_lidx = q['lidx'].data_ptr()
            args.patch_val = q['val'].data_ptr()
            args.patch_rows = q['rows']
            args.patch_umax = q['umax']
            args.patch_emax = q['emax']
            args.patch_row_bytes = q['row_bytes']
            args.n_patches = q['n']
            if q.get('ell_base') is not None:
                args.patch_ell_base = q['ell_base'].data_ptr()
        elif whole:
            # (a stored order permutes the whole row range: partial ranges
            # run without it, and without the schedules built on it)
            args.row_order = order.data_ptr() if order is not None else None

            def same_order(sched):
                return (order is None) == (sched['order'] is None) and (
                    order is None or
                    order.data_ptr() == sched['order'].data_ptr())
            patches = self.patches
            if patches is not None and same_order(patches):
                args.patch_ptr = patches['ptr'].data_ptr()
                args.patch_ucol = patches['ucol'].data_ptr()
                args.patch_rowptr = patches['rowptr'].data_ptr()
                args.patch_lidx = patches['lidx'].data_ptr()
                args.patch_val = patches['val'].data_ptr()
                args.patch_rows = patches['rows']
                args.patch_umax = patches['umax']
                args.patch_emax = patches['emax']
                args.patch_row_bytes = patches['row_bytes']
                args.n_patches = patches['n']
            groups = self.groups
            if groups is not None and same_order(groups):
                args.group_meta = groups['meta'].data_ptr()
                args.group_col = groups['col'].data_ptr()
                args.group_w = groups['w'].data_ptr()
                args.group_mask = groups['mask'].data_ptr()
                args.group_rid = groups['rid'].data_ptr()
                args.group_frac = groups['frac'].data_ptr()
                args.n_groups = groups['n']
                args.group_rows = groups['rows']
                share = groups.get('share')
                if share is not None:
                    args.share_meta = share['meta'].data_ptr()
                    args.share_col = share['col'].data_ptr()
                    args.share_mask = share['mask'].data_ptr()
                    args.share_waves = share['waves']
            if self.strips is not None:
                # (self-contained: its own row order and row ids)
                args.strips = ctypes.addressof(self.strips['struct'])
        self._args_cache[key] = (self._sched_version, bytes(args))
        return args

    # -- construction -------------------------------------------------------
    @classmethod
    def from_triplets(cls, row, col, S, frac_b, n_a, n_b, index_base=1,
                      device=None):
        """
        ``csr_matrix((S, (row - base, col - base)), shape=(n_b, n_a))`` on
        the device (``remap_numpy.py:134-137``).  ``row``/``col``/``S`` may
        be numpy arrays or torch tensors (host or device).
        """
        torch = require_gpu()
        lib = load_library()
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        device = torch.device(device)
        if n_a >= 2 ** 31 - 1 or n_b >= 2 ** 31 - 1:
            raise ValueError('the engine indexes cells with 32 bits: n_a and '
                             'n_b must stay below 2**31 - 1')

        def index32(v, name, extent):
            # Range-check BEFORE narrowing to int32: an int64 index of 2**31
            # or more would otherwise wrap into the valid range.
            t = torch.as_tensor(np.ascontiguousarray(v) if isinstance(
                v, np.ndarray) else v)
            if t.dtype != torch.int32 and t.numel():
                lo, hi = int(t.min()), int(t.max())
                if lo < index_base or hi >= extent + index_base:
                    bad = int(((t < index_base) |
                               (t >= extent + index_base)).sum())
                    raise ValueError(
                        f'{bad} mapping triplets have a {name} index outside '
                        f'[{index_base}, {extent + index_base - 1}]')
            return t.to(device=device, dtype=torch.int32)

        row_d = index32(row, 'row', n_b)
        col_d = index32(col, 'col', n_a)
        s_d = torch.as_tensor(np.ascontiguousarray(S) if isinstance(
            S, np.ndarray) else S).to(device=device, dtype=torch.float64)
        row_d, col_d, s_d = (row_d.contiguous(), col_d.contiguous(),
                             s_d.contiguous())
        nnz = int(s_d.shape[0])
        if row_d.shape[0] != nnz or col_d.shape[0] != nnz:
            raise ValueError('row, col and S must have the same length')
        frac_d = torch.as_tensor(
            np.ascontiguousarray(frac_b) if isinstance(frac_b, np.ndarray)
            else frac_b).to(device=device, dtype=torch.float64).contiguous()
        if frac_d.shape[0] != n_b:
            raise ValueError(f'frac_b has {frac_d.shape[0]} entries, '
                             f'expected n_b = {n_b}')
        with torch.cuda.device(device):
            nbytes = ctypes.c_size_t(0)
            _check(lib.remap_csr_from_coo_workspace(
                nnz, n_b, ctypes.byref(nbytes)),
                'remap_csr_from_coo_workspace')
            ws = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8,
                             device=device)
            rowptr = torch.empty(n_b + 1, dtype=torch.int64, device=device)
            col_out = torch.empty(max(nnz, 1), dtype=torch.int32,
                                  device=device)
            val_out = torch.empty(max(nnz, 1), dtype=torch.float64,
                                  device=device)
            counts = torch.zeros(2, dtype=torch.int64, device=device)
            _check(lib.remap_csr_from_coo(
                n_b, n_a, nnz, _ptr(row_d), _ptr(col_d), _ptr(s_d),
                int(index_base), _ptr(rowptr), _ptr(col_out), _ptr(val_out),
                ctypes.c_void_p(counts.data_ptr()),
                ctypes.c_void_p(counts.data_ptr() + 8), _ptr(ws),
                ws.numel(), _stream_ptr(device)), 'remap_csr_from_coo')
            nnz_u, bad = (int(v) for v in counts.cpu())
        if bad:
            raise ValueError(
                f'{bad} mapping triplets have a row or col index outside '
                f'the ({n_b}, {n_a}) matrix')
        return cls(n_a, n_b, rowptr, col_out[:nnz_u].clone(),
                   val_out[:nnz_u].clone(), frac_d)

    @classmethod
    def from_csr(cls, indptr, indices, data, frac_b, n_a, device=None):
        """
        Take an existing (host or device) CSR.  The kernels and the schedule
        builders rely on the canonical form scipy produces (``rowptr``
        monotone from 0 to ``len(indices)``, columns in range, ascending and
        unique within a row), so the input is validated and then rebuilt
        through the same device COO -> CSR path as a mapping file: unsorted
        rows are sorted and duplicate entries summed, as
        ``csr_matrix.sum_duplicates`` would.
        """
        torch = require_gpu()
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        rowptr = torch.as_tensor(indptr).to(device=device, dtype=torch.int64)
        n_b = int(rowptr.shape[0]) - 1
        col = torch.as_tensor(indices).to(device=device)
        val = torch.as_tensor(data).to(device=device, dtype=torch.float64)
        if n_b < 0 or col.shape[0] != val.shape[0]:
            raise ValueError('indptr must hold n_b + 1 entries and indices, '
                             'data the same number of entries')
        if n_b >= 0 and rowptr.numel():
            lens = rowptr[1:] - rowptr[:-1]
            if int(rowptr[0]) != 0 or int(rowptr[-1]) != col.shape[0] or \
                    (lens.numel() and int(lens.min()) < 0):
                raise ValueError(
                    'indptr must rise monotonically from 0 to len(indices)')
        else:
            lens = rowptr[:0]
        row = torch.repeat_interleave(
            torch.arange(n_b, device=device, dtype=torch.int32), lens)
        return cls.from_triplets(row, col, val, frac_b, n_a, n_b,
                                 index_base=0, device=device)

    # -- sharding -----------------------------------------------------------
    def shard_bounds(self, world_size):
        """
        Contiguous destination-row ranges, balanced by the work per row
        (entries + a constant per row for the output write), as a list of
        ``world_size + 1`` row indices.
        """
        from pyremap_amd.parallel import row_shard_bounds
        return row_shard_bounds(self.rowptr, world_size)

    def shard(self, rank, world_size):
        """The plan of rank ``rank`` of ``world_size`` (rows only)."""
        bounds = self.shard_bounds(world_size)
        return self.row_slice(bounds[rank], bounds[rank + 1])

    def row_slice(self, r0, r1):
        j0 = int(self.rowptr[r0])
        j1 = int(self.rowptr[r1])
        return RemapPlan(
            self.n_a, r1 - r0, (self.rowptr[r0:r1 + 1] - j0).contiguous(),
            self.col[j0:j1].contiguous(), self.val[j0:j1].contiguous(),
            self.frac_b[r0:r1].contiguous(),
            row_offset=self.row_offset + r0, n_b_global=self.n_b_global)

    def to(self, device):
        """A copy of this plan (CSR + frac_b, no schedule) on ``device``."""
        torch = _torch()
        device = torch.device(device)
        return RemapPlan(
            self.n_a, self.n_b, self.rowptr.to(device),
            self._col_padded.to(device), self._val_padded.to(device),
            self.frac_b.to(device), row_offset=self.row_offset,
            n_b_global=self.n_b_global)

    def packed(self):
        """
        This plan over the COMPACT space of the source rows it references
        (a row shard needs ``X[unique(col[shard])]``
        and nothing else): returns ``(plan, ucols)`` where ``ucols`` is the
        ascending int32 device tensor of the distinct source rows and
        ``plan`` is the same matrix with ``n_a = len(ucols)`` and every
        column index replaced by its position in ``ucols``
        (``remap_pack_columns``).  The renumbering is monotone, so a row's
        entries keep their order and the packed plan applied to
        ``X[ucols]`` gives the same bits as this plan applied to ``X``.
        Schedules are not carried over (they hold column indices): call
        ``auto_schedule`` on the result.
        """
        torch = _torch()
        lib = load_library()
        dev = self.device
        with torch.cuda.device(dev):
            nbytes = ctypes.c_size_t(0)
            _check(lib.remap_pack_columns_workspace(
                self.n_a, ctypes.byref(nbytes)),
                'remap_pack_columns_workspace')
            ws = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8,
                             device=dev)
            col = torch.zeros(self.nnz + CSR_PAD, dtype=torch.int32,
                              device=dev)
            ucols = torch.empty(max(min(self.nnz, self.n_a), 1),
                                dtype=torch.int32, device=dev)
            counts = torch.zeros(2, dtype=torch.int64, device=dev)
            _check(lib.remap_pack_columns(
                _ptr(self.col), self.nnz, self.n_a, _ptr(col), _ptr(ucols),
                ctypes.c_void_p(counts.data_ptr()),
                ctypes.c_void_p(counts.data_ptr() + 8), _ptr(ws),
                ws.numel(), _stream_ptr(dev)), 'remap_pack_columns')
            n_u, bad = (int(v) for v in counts.cpu())
        if bad:
            raise EngineError(f'{bad} column indices outside [0, {self.n_a})')
        plan = RemapPlan(n_u, self.n_b, self.rowptr, col, self._val_padded,
                         self.frac_b, row_offset=self.row_offset,
                         n_b_global=self.n_b_global)
        return plan, ucols[:n_u].clone()

    # -- scheduling ---------------------------------------------------------
    def set_row_order(self, order):
        """
        Install (or clear, with ``None``) the processing order of the rows:
        a permutation of ``range(n_b)``.  Results never depend on it.
        """
        torch = _torch()
        if order is None:
            self.row_order = None
            return
        order = torch.as_tensor(order).to(device=self.device,
                                          dtype=torch.int32).contiguous()
        if order.shape != (self.n_b,):
            raise ValueError(f'row order must have {self.n_b} entries')
        self.row_order = order

    def set_grid_schedule(self, grid_dims, kind='tile', tile=(32, 64)):
        """
        Walk a 2-D destination grid (``grid_dims`` = C-order dims of the
        WHOLE mapping's destination) in tiles or along a Morton curve, so
        rows that are neighbours in either direction -- and therefore share
        source rows -- are computed close together in time and on the same
        XCD.  1-D destinations keep their natural order.
        """
        torch = _torch()
        if grid_dims is None or len(grid_dims) != 2 or kind in (None, 'none'):
            self.row_order = None
            return
        my, mx = (int(d) for d in grid_dims)
        if my * mx != self.n_b_global:
            raise ValueError(f'grid {grid_dims} does not hold '
                             f'{self.n_b_global} cells')
        rows = torch.arange(self.row_offset, self.row_offset + self.n_b,
                            device=self.device, dtype=torch.int64)
        jy = rows // mx
        jx = rows - jy * mx
        if kind == 'tile':
            ty, tx = (int(t) for t in tile)
            ntx = (mx + tx - 1) // tx
            key = ((jy // ty) * ntx + jx // tx) * (ty * tx) + \
                (jy % ty) * tx + jx % tx
        elif kind == 'morton':
            key = torch.zeros_like(rows)
            for bit in range(16):
                key |= ((jx >> bit) & 1) << (2 * bit)
                key |= ((jy >> bit) & 1) << (2 * bit + 1)
        else:
            raise ValueError(f'unknown schedule {kind!r}')
        self.row_order = torch.argsort(key, stable=True).to(torch.int32)

    def _csr_struct(self):
        csr = _CSR()
        csr.n_rows, csr.n_cols, csr.nnz = self.n_b, self.n_a, self.nnz
        csr.rowptr = self.rowptr.data_ptr()
        csr.col = self.col.data_ptr()
        csr.val = self.val.data_ptr()
        csr.max_row_nnz, csr.csr_pad = self.max_row_nnz, self.csr_pad
        return csr

    def build_patches(self, grid_dims=None, tile=(4, 8), lds_budget=80 * 1024,
                      row_bytes=1024):
        """
        Build the LDS-staging schedule (``remap_apply_args.patch_*``) with
        the library's device builder (``remap_patches_build``): destination
        rows are walked in ``tile`` order over a 2-D grid (natural order for
        1-D destinations), ``tile[0] * tile[1]`` consecutive work slots form
        a patch, and for every patch the distinct source rows are listed
        once.  The tile is halved until the longest list fits the LDS budget
        (``row_bytes`` per staged source-row chunk).  Returns the fraction
        distinct / entries (small = much reuse), or ``None`` if no patch size
        fits.
        """
        def fits(rows, umax, emax):
            # entries staged next to the rows: 12 B each (+24 B per row)
            return (umax + 1) * row_bytes + emax * 12 + rows * 24 + 32 <= \
                lds_budget
        q = self._make_patches(grid_dims, tile, fits, row_bytes)
        if q is None:
            self.patches = None
            if self.nnz and self.n_b:
                self.row_order = None
            return None
        self.row_order = q['order']
        self.patches = q
        return q['distinct'] / self.nnz

    def _make_patches(self, grid_dims, tile, fits, row_bytes):
        """``remap_patches_build`` with ``tile``, halved until ``fits(rows,
        umax, emax)``; returns the plan's arrays as a dict, or ``None``."""
        torch = _torch()
        lib = load_library()
        if self.nnz == 0 or self.n_b == 0:
            return None
        dev = self.device
        ty, tx = (int(t) for t in tile)
        two_d = grid_dims is not None and len(grid_dims) == 2
        dims = None
        if two_d:
            my, mx = (int(d) for d in grid_dims)
            if my * mx != self.n_b_global:
                raise ValueError(f'grid {grid_dims} does not hold '
                                 f'{self.n_b_global} cells')
            dims = (ctypes.c_int64 * 2)(my, mx)
        csr = self._csr_struct()
        with torch.cuda.device(dev):
            nbytes = ctypes.c_size_t(0)
            _check(lib.remap_patches_workspace(self.n_b, self.nnz,
                                               ctypes.byref(nbytes)),
                   'remap_patches_workspace')
            ws = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8,
                             device=dev)
            ucol = torch.empty(self.nnz, dtype=torch.int32, device=dev)
            prow = torch.empty(self.n_b + 1, dtype=torch.int32, device=dev)
            lidx = torch.empty(self.nnz, dtype=torch.int32, device=dev)
            pval = torch.empty(self.nnz, dtype=torch.float64, device=dev)
            order = torch.empty(self.n_b, dtype=torch.int32, device=dev) \
                if two_d else None
            stats = torch.zeros(3, dtype=torch.int64, device=dev)
            while True:
                rows = ty * tx
                n_patches = (self.n_b + rows - 1) // rows
                ptr = torch.empty(n_patches + 1, dtype=torch.int32,
                                  device=dev)
                _check(lib.remap_patches_build(
                    ctypes.byref(csr), dims, self.row_offset, ty, tx,
                    _ptr(order), _ptr(ptr), _ptr(ucol), _ptr(prow),
                    _ptr(lidx), _ptr(pval), _ptr(stats), _ptr(ws),
                    ws.numel(), _stream_ptr(dev)), 'remap_patches_build')
                distinct, umax, emax = (int(v) for v in stats.cpu())
                if fits(rows, umax, emax):
                    break
                if rows == 1:
                    return None
                if tx >= ty and tx > 1:
                    tx //= 2
                else:
                    ty //= 2
        return dict(
            ptr=ptr, ucol=ucol[:max(distinct, 1)].clone(), rowptr=prow,
            lidx=lidx, val=pval, rows=rows, umax=umax, emax=emax,
            n=n_patches, order=order, tile=(ty, tx), distinct=distinct,
            row_bytes=int(row_bytes))

    #: distinct source cells a patch of the lanes-across-rows kernel may
    #: stage: two cells per lane of a 1 024-thread workgroup, two LDS images
    #: of 4 fields each = 128 KB (spmm_patchtime)
    CELL_UMAX = 2046

    def run_patches(self):
        """
        Small LDS patches (4 x 8 tiles of the destination grid, 32
        consecutive rows of a 1-D one) for ``(Time, nCells, L)`` fields with
        SHORT level runs, 4 <= L < 16, on mappings whose own schedule is the
        row groups: there every (source row, batch) is a separate 32-120 byte
        run, the row-group kernel pays a line fetch per run and ENTRY, the
        LDS patch kernel (family 5) one per run and DISTINCT source row
        (config 3's map, fraction of 8 TB/s, groups / patches: L = 4 0.19 /
        0.27, 6 0.19 / 0.27, 8 0.39 / 0.45, 10 0.28 / 0.38, 12 0.34 / 0.41;
        from L = 16 the groups win, 0.60 / 0.55).  Built on first use;
        ``None`` when there is nothing to build.
        """
        if self._runs is None:
            dims = self._grid_dims
            if dims is not None and len(dims) != 2:
                dims = None

            def fits(rows, umax, emax):
                return (umax + 1) * 1024 + emax * 12 + rows * 24 + 32 <= \
                    100 * 1024 or rows <= 4
            q = self._make_patches(dims, (4, 8) if dims is not None
                                   else (1, 32), fits, 1024)
            self._runs = q if q is not None else False
            self._sched_version += 1
        return self._runs or None

    def run_cells(self):
        """
        The patch plan of the batch-at-a-time lanes-across-rows kernel
        (``spmm_patchtime<..., RUNS>``) for ``(Time, nCells, L)`` fields with
        VERY short level runs, 4 <= L <= ``RUN_CELLS_MAX``: 16 x 16 tiles (256
        consecutive rows of a 1-D destination), halved until a patch holds at
        most 510 source cells -- one 256-thread workgroup per patch walks the
        batches, the results of a batch leave through LDS as whole lines
        (config 3's map, ms per launch, this kernel / the small LDS patches
        of :meth:`run_patches` / row groups: L = 4 0.58 / 0.79 / 1.11, 5
        0.70 / 0.99 / 1.54, 6 0.74 / 0.77 / 1.12; from L = 8 the LDS patches
        win: 0.68 / 0.53 / 0.61).  Built on first use.
        """
        if self._run_cells is None:
            dims = self._grid_dims
            if dims is not None and len(dims) != 2:
                dims = None
            q = self._make_patches(
                dims, (16, 16) if dims is not None else (1, 256),
                lambda rows, umax, emax: umax <= 510 or rows <= 16, 1024)
            self._run_cells = q if q is not None else False
            self._sched_version += 1
        return self._run_cells or None

    #: longest level run the batch-at-a-time kernel takes (see run_cells)
    RUN_CELLS_MAX = 6

    def cell_patches(self):
        """
        The patch plan of kernel family 7 (``spmm_patchcell``), which serves
        fields whose contiguous run behind the source axes is short --
        (Time, nCells), the reference's most common input -- built on first
        use: 16 x 16 tiles of the destination grid (256 consecutive rows of a
        1-D destination), halved until no patch references more than
        ``CELL_UMAX`` distinct source cells.  ``None`` when there is nothing
        to build.
        """
        if self._cell is None:
            dims = self._grid_dims
            if dims is not None and len(dims) != 2:
                dims = None
            # (small grids: smaller tiles, so that there still are a few
            # hundred workgroups)
            tile = self.CELL_TILE
            while tile[0] * tile[1] > 256 and \
                    self.n_b < 128 * tile[0] * tile[1]:
                tile = (tile[0], tile[1] // 2) if tile[1] >= tile[0] else \
                    (tile[0] // 2, tile[1])
            def build(tile):
                return self._make_patches(
                    dims, tile if dims is not None
                    else (1, tile[0] * tile[1]),
                    lambda rows, umax, emax: umax <= self.CELL_UMAX or
                    rows <= 16, 1024)
            q = build(tile)
            # coarse -> fine (a patch stages fewer than half as many source
            # cells as it has rows): the output stores are the work, and
            # four 256-thread workgroups do them sooner than one of 1 024 --
            # 1 deg -> 0.5 deg bilinear, short rows, (12, n) 20.3 -> 17.3 us,
            # (120, n) float32 75.8 -> 71.9
            # (only where the large patches are few -- 254 there; config 4's
            # map, 29 K patches of 32 x 32, loses with the small ones: (32, n)
            # 3.3 -> 5.7 ms)
            if q is not None and q['rows'] > 256 and \
                    2 * q['umax'] < q['rows'] and q['n'] < 1024:
                ty, tx = q['tile']
                while ty * tx > 256:
                    ty, tx = (ty, tx // 2) if tx >= ty else (ty // 2, tx)
                q = build((ty, tx))
            self._cell = q if q is not None else False
            self._sched_version += 1
        return self._cell or None

    #: tile of the destination grid a patch of cell_patches covers (halved
    #: until a patch holds at most CELL_UMAX source cells): one 1 024-thread
    #: workgroup per patch and CU walks ALL its chunks -- config 3's map,
    #: (120, nCells) cold: 16 x 16 / 16 x 32 / 32 x 32 tiles 0.146 / 0.137 /
    #: 0.121 ms
    CELL_TILE = (32, 32)

    GROUP = 8   # default rows per group (remap_apply_args.group_rows)

    def build_groups(self, grid_dims=None, super_tile=32, rows=None,
                     share=0):
        """
        Build the row-group schedule (``remap_apply_args.group_*``) with the
        library's device builder (``remap_groups_build``): ``rows`` (8 or 4)
        consecutive work slots -- a 2 x 4 or 2 x 2 tile of a 2-D destination
        grid, walked row-major inside ``super_tile`` x ``super_tile`` blocks
        -- share one sorted list of the distinct source rows they reference;
        the weights are stored for the present (union entry, member) pairs
        only, in that order.  Returns union entries / entries (small = many
        shared source rows).

        ``share`` = 2 or 4: also build the SHARED union lists of the shared
        form (``csrc/spmm_groupshare.h``, ``remap_share_build``): 2 / 4
        consecutive 8-row groups -- a 4 x 4 / 4 x 8 tile of the grid, walked
        row-major inside the supertiles -- get one union served by one
        workgroup through LDS.  ``groups['share']['ratio']`` is their union
        entries / entries.
        """
        torch = _torch()
        lib = load_library()
        G = int(rows or self.GROUP)
        if G not in (4, 8, 16):
            raise ValueError('row groups hold 4, 8 or 16 rows')
        share = int(share or 0)
        if share:
            if share not in (2, 4) or G != 8:
                raise ValueError('shared lists: 2 or 4 groups of 8 rows')
            if int(super_tile) < (1 << 30) and int(super_tile) % (2 * share):
                raise ValueError('super_tile must hold whole 4 x '
                                 f'{2 * share} supergroup tiles')
        self.groups = None
        if self.nnz == 0 or self.n_b == 0:
            return None
        dev = self.device
        two_d = grid_dims is not None and len(grid_dims) == 2
        dims = None
        if two_d:
            my, mx = (int(d) for d in grid_dims)
            if my * mx != self.n_b_global:
                raise ValueError(f'grid {grid_dims} does not hold '
                                 f'{self.n_b_global} cells')
            dims = (ctypes.c_int64 * 2)(my, mx)
        st = int(super_tile)
        if st >= 1 << 30:
            st = 0           # no supertiles: row-major over the whole grid
        n_groups = (self.n_b + G - 1) // G
        meta = torch.empty((n_groups + 1, 2), dtype=torch.int64, device=dev)
        col = torch.empty(self.nnz + 32, dtype=torch.int32, device=dev)
        mask = torch.empty(self.nnz + 32, dtype=torch.int32, device=dev)
        w = torch.empty(self.nnz + 128, dtype=torch.float64, device=dev)
        rid = torch.empty(n_groups * G, dtype=torch.int32, device=dev)
        frac = torch.empty(n_groups * G, dtype=torch.float64, device=dev)
        order = torch.empty(self.n_b, dtype=torch.int32, device=dev) \
            if two_d else None
        n_union = torch.zeros(1, dtype=torch.int64, device=dev)
        csr = self._csr_struct()
        with torch.cuda.device(dev):
            nbytes = ctypes.c_size_t(0)
            _check(lib.remap_groups_workspace(self.n_b, self.nnz,
                                              ctypes.byref(nbytes)),
                   'remap_groups_workspace')
            ws = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8,
                             device=dev)
            _check(lib.remap_groups_build(
                ctypes.byref(csr), _ptr(self.frac_b), G, dims,
                self.row_offset, st, share, _ptr(order), _ptr(meta),
                _ptr(col),
                _ptr(mask), _ptr(w), _ptr(rid), _ptr(frac), _ptr(n_union),
                _ptr(ws), ws.numel(), _stream_ptr(dev)),
                'remap_groups_build')
            nu = int(n_union)
        self.row_order = order
        # trim the union arrays to what is used (+ the readable pad)
        groups = dict(meta=meta, col=col[:nu + 32].clone(),
                      w=w, mask=mask[:nu + 32].clone(), rid=rid,
                      frac=frac, n=n_groups, rows=G, order=order,
                      union=nu)
        if share:
            n_super = (self.n_b + 8 * share - 1) // (8 * share)
            smeta = torch.empty((n_super + 1, 2), dtype=torch.int64,
                                device=dev)
            # (col / mask above are free again: trimmed copies were taken)
            scol = torch.empty(self.nnz + 256, dtype=torch.int32, device=dev)
            smask = torch.empty(self.nnz + 256, dtype=torch.int32,
                                device=dev)
            with torch.cuda.device(dev):
                _check(lib.remap_share_build(
                    ctypes.byref(csr), _ptr(rid), share, _ptr(smeta),
                    _ptr(scol), _ptr(smask), _ptr(n_union), _ptr(ws),
                    ws.numel(), _stream_ptr(dev)), 'remap_share_build')
                su = int(n_union)
            groups['share'] = dict(meta=smeta, col=scol[:su + 256].clone(),
                                   mask=smask[:su + 256].clone(),
                                   waves=share, union=su,
                                   ratio=su / self.nnz)
        self.groups = groups
        return nu / self.nnz

    #: a row counts as LONG from this many entries on (the widest stencils
    #: of ordinary maps -- 2nd-order conservative -- hold ~30)
    LONG_ROW = 96

    def build_strips(self, grid_dims, **shape):
        """
        Attach the strip schedule of kernel family 8 (``csrc/spmm_strip.h``):
        an LDS ring of source-row pieces sliding along strips of the
        destination grid ``grid_dims``, for entry-rich mappings (2nd-order
        conservative stencils).  ``shape``: ``strip_rows``, ``step_cols``,
        ``segments``, ``depth`` (:func:`pyremap_amd.strips.build_strips`).
        Calls the kernel cannot serve (several batches, float32, a row
        shard) run on the plan's other schedule as before.  Raises
        :class:`pyremap_amd.strips.StripsUnfit` when the ring does not fit
        the LDS.
        """
        from pyremap_amd import strips as _strips
        if self.row_offset != 0 or self.n_b != self.n_b_global:
            raise ValueError('a strip schedule covers a whole mapping')
        st = _strips.build_strips(self, grid_dims, **shape)
        st['struct'] = _strips.struct(st, _Strips)
        self.strips = st
        return st

    def _split_long_rows(self):
        """
        Mappings whose few LONG rows hold a large share of the entries -- a
        global lat-lon bilinear map as ESMF makes it: the destination cells
        poleward of the last source row take the WHOLE adjacent source row
        (the pole cap, ``weights.bilinear_3d``), 362-1 442 entries per row
        among rows of 4, a third of all entries in 0.6 % of the rows -- lose
        every schedule built for short rows (no LDS patch holds 1 440 source
        rows) and serialise the others (a lane walks its row entry by entry:
        1 deg -> 0.5 deg, one 2-D field, 66 us against 13 us without the caps;
        K = 64: 150 against 37).

        Such a plan is applied as TWO launches writing disjoint rows: the
        mapping without the long rows' entries on its own schedule, and the
        long rows through the LDS-staged lanes-across-rows kernel (family 7):
        the source rows they share are staged once per 256 rows, every lane
        sums its row in CSR order from LDS.  Returns ``(short, long)`` plans
        or ``None`` when the mapping has no such rows (or the long rows'
        sources do not fit the LDS).
        """
        torch = _torch()
        if self.max_row_nnz <= self.LONG_ROW:
            return None
        counts = self.rowptr[1:] - self.rowptr[:-1]
        is_long = counts > self.LONG_ROW
        ids = is_long.nonzero().squeeze(1)
        n_long = int(ids.shape[0])
        long_entries = int(counts[ids].sum())
        if n_long == 0 or n_long > self.n_b // 8 or \
                long_entries < self.nnz // 50:
            return None
        entry_long = torch.repeat_interleave(is_long, counts)

        def sub(rows_kept, entries_kept, n_rows, **where):
            rp = torch.zeros(n_rows + 1, dtype=torch.int64,
                             device=self.device)
            rp[1:] = torch.cumsum(rows_kept, 0)
            return RemapPlan(self.n_a, n_rows, rp,
                             self.col[entries_kept].contiguous(),
                             self.val[entries_kept].contiguous(),
                             self.frac_b, **where)
        # (a row shard splits like a whole mapping: same place in the grid)
        short = sub(torch.where(is_long, torch.zeros_like(counts), counts),
                    ~entry_long, self.n_b, row_offset=self.row_offset,
                    n_b_global=self.n_b_global)
        long = sub(counts[ids], entry_long, n_long)
        # the long rows' patch plan: 256 consecutive long rows per workgroup,
        # halved until the distinct source rows fit the LDS with 4 fields
        # per lane; work slot -> row of the WHOLE mapping
        limit = CELL_LDS_MAX // (4 * 8) - 2
        q = long._make_patches(None, (1, 256),
                               lambda rows, umax, emax: umax <= limit, 1024)
        if q is None:
            return None
        q['order'] = ids.to(torch.int32).contiguous()
        # many fields: a wave per long row, the source cells of a few
        # neighbouring rows sliding through LDS (family 11) -- row-major
        # entries whose local indices do not decrease inside a row (they
        # cannot: the rows are sorted by column; checked all the same)
        long._wave = None
        # (its LDS image per row: two windows of 8 cells x 512 bytes and the
        # row's records, 12 bytes each -- run_longwave)
        per_row = 2 * 8 * 512 + ((long.max_row_nnz + 15) // 16 * 16 + 16) * 12
        wave_rows = min(int(LONG_WAVE_ROWS), CELL_LDS_MAX // per_row)
        if wave_rows >= 1:
            w = long._make_patches(None, (1, wave_rows),
                                   lambda rows, umax, emax: True, 1024)
            if w is not None:
                li = w['lidx'][:long.nnz]
                first = torch.zeros(long.nnz, dtype=torch.bool,
                                    device=self.device)
                first[w['rowptr'][:-1].to(torch.int64)
                      .clamp_(max=long.nnz - 1)] = True
                if bool(((li[1:] >= li[:-1]) | first[1:]).all()):
                    w['order'] = q['order']
                    long._wave = w
        rows, n_p = q['rows'], q['n']
        rp = q['rowptr'].to(torch.int64)
        lens = rp[1:] - rp[:-1]
        slot = torch.arange(n_long, device=self.device)
        patch, r_local = slot // rows, slot % rows
        e_slot = torch.repeat_interleave(slot, lens)
        j = torch.arange(long.nnz, device=self.device) - rp[e_slot]
        lidx, val = q['lidx'][:long.nnz], q['val'][:long.nnz]
        # the entries COLUMN-MAJOR inside every patch (patch_ell_base):
        # the lanes -- one slot each -- read their j-th entries with one
        # coalesced load instead of a line fetch per lane and entry
        longest = torch.zeros(n_p, dtype=torch.int64,
                              device=self.device) \
            .scatter_reduce(0, patch, lens, 'amax')
        base = torch.cumsum(longest * rows, 0) - longest * rows
        dest = base[patch[e_slot]] + j * rows + r_local[e_slot]
        total = int((longest * rows).sum())
        val_t = torch.zeros(total + CSR_PAD, dtype=torch.float64,
                            device=self.device)
        lidx_t = torch.zeros(total + CSR_PAD, dtype=torch.int32,
                             device=self.device)
        val_t[dest] = val
        lidx_t[dest] = lidx
        q['val'], q['lidx'] = val_t, lidx_t
        q['ell_base'] = base.contiguous()
        q['layout'] = 'column-major'
        long._cell = q
        return short, long

    def auto_schedule(self, grid_dims, _split_ok=True):
        """
        Choose AND build the schedule for this mapping with the library's
        ``remap_schedule_auto`` (the rules -- LDS patches for heavily shared,
        short rows; row groups where rows share columns at all; the plain
        kernel otherwise -- live there, next to the measurements they come
        from: ``csrc/remap_schedule.hip``).  The
        schedule's arrays are views into one device arena owned by the plan.
        Returns the description of what was chosen.
        """
        torch = _torch()
        lib = load_library()
        self.patches = None
        self.groups = None
        self.row_order = None
        self.default_tune = None
        self._arena = None
        self._cell = None
        self._runs = None
        self._run_cells = None
        self._grid_dims = None
        self._split = None
        if grid_dims is None or self.nnz == 0 or self.n_b == 0:
            return {'family': 'rowscalar', 'reason': 'no destination grid'}
        if _split_ok:
            split = self._split_long_rows()
            if split is not None:
                short, long = split
                choice = short.auto_schedule(grid_dims, _split_ok=False)
                self._split = split
                self._grid_dims = short._grid_dims
                return dict(choice, long_rows=long.n_b,
                            long_row_entries=long.nnz,
                            long_rows_layout=long._cell['layout'])
        dims = tuple(int(d) for d in grid_dims)
        if len(dims) in (1, 2) and _prod(dims) == self.n_b_global:
            self._grid_dims = dims
        if len(dims) not in (1, 2):
            return {'family': 'rowscalar',
                    'reason': f'{len(dims)}-D destination grid'}
        if _prod(dims) != self.n_b_global:
            raise ValueError(f'grid {dims} does not hold '
                             f'{self.n_b_global} cells')
        dev = self.device
        sched = _Schedule()
        csr = self._csr_struct()
        cdims = (ctypes.c_int64 * len(dims))(*dims)
        with torch.cuda.device(dev):
            a_bytes, w_bytes = ctypes.c_size_t(0), ctypes.c_size_t(0)
            _check(lib.remap_schedule_sizes(
                self.n_b, self.nnz, ctypes.byref(a_bytes),
                ctypes.byref(w_bytes)), 'remap_schedule_sizes')
            arena = torch.empty(int(a_bytes.value), dtype=torch.uint8,
                                device=dev)
            ws = torch.empty(max(int(w_bytes.value), 1), dtype=torch.uint8,
                             device=dev)
            _check(lib.remap_schedule_auto(
                ctypes.byref(csr), _ptr(self.frac_b), cdims, len(dims),
                self.row_offset, _ptr(arena), arena.numel(), _ptr(ws),
                ws.numel(), ctypes.byref(sched), _stream_ptr(dev)),
                'remap_schedule_auto')
            del ws
            if sched.family in (5, 10) and \
                    sched.arena_used < arena.numel() // 2:
                # keep only what the schedule occupies (the arena was sized
                # for the larger of the two candidate layouts)
                used = (int(sched.arena_used) + 255) // 256 * 256
                small = arena[:used].clone()
                shift = small.data_ptr() - arena.data_ptr()
                for name, ctype in _Schedule._fields_:
                    v = getattr(sched, name)
                    if ctype is ctypes.c_void_p and v and \
                            arena.data_ptr() <= v < arena.data_ptr() + used:
                        setattr(sched, name, v + shift)
                arena = small
        base = arena.data_ptr()

        def view(ptr, count, dtype):
            if not ptr:
                return None
            nbytes = count * torch.empty((), dtype=dtype).element_size()
            return arena[ptr - base:ptr - base + nbytes].view(dtype)

        self._arena = arena
        order = view(sched.row_order, self.n_b, torch.int32)
        tune = {mode: [int(v) for v in sched.tune[mode]][:6]
                for mode in (MODE_RAW, MODE_FRACB, MODE_MASKED)}
        if sched.family == 5:
            self.row_order = order
            n_p = int(sched.n_patches)
            self.patches = dict(
                ptr=view(sched.patch_ptr, n_p + 1, torch.int32),
                ucol=view(sched.patch_ucol, max(int(sched.n_distinct), 1),
                          torch.int32),
                rowptr=view(sched.patch_rowptr, self.n_b + 1, torch.int32),
                lidx=view(sched.patch_lidx, self.nnz, torch.int32),
                val=view(sched.patch_val, self.nnz, torch.float64),
                rows=int(sched.patch_rows), umax=int(sched.patch_umax),
                emax=int(sched.patch_emax), n=n_p, order=order,
                tile=(int(sched.tile_y), int(sched.tile_x)),
                distinct=int(sched.n_distinct),
                row_bytes=int(sched.patch_row_bytes))
            return {'family': 'patch', 'tile': self.patches['tile'],
                    'ratio': float(sched.ratio),
                    'umax': self.patches['umax'],
                    'row_bytes': self.patches['row_bytes']}
        if sched.family == 10:
            self.row_order = order
            n_g, G = int(sched.n_groups), int(sched.group_rows)
            nu = int(sched.n_distinct)
            self.groups = dict(
                meta=view(sched.group_meta, 2 * (n_g + 1),
                          torch.int64).reshape(n_g + 1, 2),
                col=view(sched.group_col, nu + 32, torch.int32),
                w=view(sched.group_w, self.nnz + 128, torch.float64),
                mask=view(sched.group_mask, nu + 32, torch.int32),
                rid=view(sched.group_rid, n_g * G, torch.int32),
                frac=view(sched.group_frac, n_g * G, torch.float64),
                n=n_g, rows=G, order=order, union=nu)
            share = None
            if sched.share_waves:
                W, su = int(sched.share_waves), int(sched.n_share_union)
                n_s = (self.n_b + 8 * W - 1) // (8 * W)
                share = dict(
                    meta=view(sched.share_meta, 2 * (n_s + 1),
                              torch.int64).reshape(n_s + 1, 2),
                    col=view(sched.share_col, su + 256, torch.int32),
                    mask=view(sched.share_mask, su + 256, torch.int32),
                    waves=W, union=su, ratio=su / self.nnz)
                self.groups['share'] = share
                # (assigning into the dict does not pass the property)
                self._sched_version += 1
            self.default_tune = tune
            rich = bool(sched.entry_rich)
            out = {'family': 'rowgroup', 'union_ratio': float(sched.ratio),
                   'rows_per_group': G,
                   'order': '2x4 groups in 32x32 supertiles' if rich
                   else '2x2 groups, row-major' if len(dims) == 2 else
                   '4 consecutive rows', 'tune': self.default_tune}
            if share is not None:
                out['shared_by'] = share['waves']
                out['shared_union_ratio'] = share['ratio']
                out['order'] = '2x4 groups in 4x8 tiles in 32x32 supertiles'
            return out
        if sched.family == 6:
            self.row_order = order.clone()
            self._arena = None
            self.default_tune = tune[MODE_FRACB]
            return {'family': 'rowscalar', 'order': 'tile 32x32',
                    'tune': self.default_tune,
                    'reason': 'entry-rich rows: keep the stencil band in L2'}
        self._arena = None
        return {'family': 'rowscalar', 'reason': 'little source-row reuse'}

    # -- accounting ---------------------------------------------------------
    def block_source_extent(self, rows_per_block):
        """
        For consecutive blocks of ``rows_per_block`` destination rows: one
        past the LAST source row any of the block's entries references, as a
        running maximum over the blocks (a list, one value per block).  A
        block can be computed once that many source rows are resident.
        Cached: the per-row maxima once per plan, the block values per size.
        """
        torch = _torch()
        rows_per_block = int(rows_per_block)
        if rows_per_block in self._extent_cache:
            return self._extent_cache[rows_per_block]
        if self._row_last_col is None:
            last = torch.zeros(self.n_b, dtype=torch.int64,
                               device=self.device)
            if self.nnz:
                # columns ascend within a row: the last entry is the largest
                lens = self.rowptr[1:] - self.rowptr[:-1]
                has = lens > 0
                last[has] = self.col[(self.rowptr[1:][has] - 1)].to(
                    torch.int64) + 1
            self._row_last_col = last
        n_blocks = (self.n_b + rows_per_block - 1) // rows_per_block
        padded = torch.zeros(n_blocks * rows_per_block, dtype=torch.int64,
                             device=self.device)
        padded[:self.n_b] = self._row_last_col
        hi = padded.reshape(n_blocks, rows_per_block).amax(dim=1)
        out = torch.cummax(hi, 0).values.cpu().tolist()
        self._extent_cache[rows_per_block] = out
        return out

    def touched_sources(self):
        """Number of DISTINCT source cells this plan's rows reference."""
        if self._touched is None:
            torch = _torch()
            hit = torch.zeros(self.n_a, dtype=torch.bool, device=self.device)
            if self.nnz:
                hit[self.col.to(torch.int64)] = True
            self._touched = int(hit.sum())
        return self._touched

    def algorithmic_bytes(self, K, x_itemsize=8, mode=MODE_FRACB):
        """
        SURVEY.md section 8(d): S + col read once, rowptr, X read once,
        Y written once, frac_b in the unmasked mode.  Only source rows that
        some entry references count towards X (a whole mapping touches every
        source cell, so this is ``n_a`` there; a row shard or a map with
        unreferenced cells moves fewer bytes and is priced accordingly).
        """
        b = self.nnz * 12 + (self.n_b + 1) * 8
        b += self.touched_sources() * K * x_itemsize + self.n_b * K * 8
        if mode == MODE_FRACB:
            b += self.n_b * 8
        return b

    def to_host_csr(self):
        return (self.rowptr.cpu().numpy(), self.col.cpu().numpy(),
                self.val.cpu().numpy())


# ---------------------------------------------------------------------------
# launches
# ---------------------------------------------------------------------------

def apply_strided(plan, X, Y, *, n_batch, k_inner, x_row_stride,
                  x_batch_stride, y_row_stride, y_batch_stride, mode,
                  threshold=0.0, mask_out=None, flags=0, tune=None,
                  row_begin=0, row_end=None, gate=None, gate_value=0,
                  x_src_fold=0, x_outer_stride=0, _long_rows=False):
    """
    One asynchronous ``remap_apply_f64`` launch on torch's current stream.
    ``X``/``Y``/``mask_out`` are device tensors; strides are in elements.

    The first ``(Time, nCells)``-like call on a plan (short contiguous runs
    in several batches) builds the patch plan of ``spmm_patchcell``
    (:meth:`RemapPlan.cell_patches`: allocations and one readback): make it
    -- or call ``plan.cell_patches()`` -- before capturing launches in a
    hipGraph.
    """
    torch = _torch()
    lib = load_library()
    if X.device != plan.device or Y.device != plan.device:
        raise ValueError('X, Y and the plan must live on the same device')
    if Y.dtype != torch.float64:
        raise TypeError('Y must be float64')
    if X.dtype == torch.float64:
        x_dtype = DTYPE_F64
    elif X.dtype == torch.float32:
        x_dtype = DTYPE_F32
    else:
        raise TypeError(f'X must be float64 or float32, not {X.dtype}')
    # the kernel addresses through raw pointers: the tensors must hold the
    # last element the strides reach
    for name, t, rows, rs, bs in (('X', X, plan.n_a, x_row_stride,
                                   x_batch_stride),
                                  ('Y', Y, plan.n_b, y_row_stride,
                                   y_batch_stride),
                                  ('mask_out', mask_out, plan.n_b,
                                   y_row_stride, y_batch_stride)):
        if t is None or n_batch <= 0 or k_inner <= 0 or rows <= 0:
            continue
        reach = (n_batch - 1) * bs + (rows - 1) * rs + k_inner
        if name == 'X' and x_src_fold:
            # two source axes: the last cell is (n_a / fold - 1, fold - 1)
            if x_src_fold < 0 or plan.n_a % x_src_fold or \
                    x_outer_stride < 0:
                raise ValueError(f'x_src_fold {x_src_fold} does not divide '
                                 f'n_a = {plan.n_a}')
            reach = (n_batch - 1) * bs + \
                (plan.n_a // x_src_fold - 1) * x_outer_stride + \
                (x_src_fold - 1) * rs + k_inner
        if min(rs, bs) < 0 or reach > t.numel():
            raise ValueError(
                f'{name} holds {t.numel()} elements; the strides (row {rs}, '
                f'batch {bs}) reach {reach}')
    if mask_out is not None and (mask_out.dtype != torch.uint8 or
                                 mask_out.device != plan.device):
        raise TypeError('mask_out must be a uint8 tensor on the plan device')
    end = plan.n_b if row_end is None else row_end
    whole = row_begin == 0 and end == plan.n_b
    if plan._split is not None and whole and not tune:
        # a few long rows hold a large share of the entries (pole caps of a
        # global bilinear map): two launches writing disjoint rows
        # (RemapPlan._split_long_rows)
        kw = dict(n_batch=n_batch, k_inner=k_inner,
                  x_row_stride=x_row_stride, x_batch_stride=x_batch_stride,
                  y_row_stride=y_row_stride, y_batch_stride=y_batch_stride,
                  mode=mode, threshold=threshold, mask_out=mask_out,
                  flags=flags, gate=gate, gate_value=gate_value,
                  x_src_fold=x_src_fold, x_outer_stride=x_outer_stride)
        short, long = plan._split
        # (Issuing the two launches on two streams joined by events -- they
        # write disjoint rows -- was built and measured: replayed from a
        # hipGraph the two branches still ran one after the other, K = 64
        # 63.5 us against 60.2; issued from Python the four extra stream
        # calls cost more than the overlap gave, 84 us against 61.  One
        # stream it stays.)
        for part, is_long in ((short, False), (long, True)):
            if part is None:      # (tools/long_rows_probe.py: one part alone)
                continue
            apply_strided(part, X, Y, _long_rows=is_long, **kw)
        return
    # short contiguous runs in several batches -- (Time, nCells) -- go to the
    # LDS-staged lanes-across-rows kernel on its own patch plan
    cell = _long_rows or (
        whole and not tune and
        ((k_inner < CELL_MAX_RUN and n_batch > 1) or x_src_fold) and
        n_batch * k_inner >= 2 and plan.cell_patches() is not None)
    # short level runs in several batches -- (Time, nCells, 4 ... 15) -- on
    # a row-group mapping: small LDS patches (RemapPlan.run_patches)
    if not cell and whole and not tune and n_batch > 1 and \
            4 <= k_inner < 16 and n_batch * k_inner >= 64 and \
            plan.patches is None:
        if k_inner <= plan.RUN_CELLS_MAX and plan.run_cells() is not None:
            cell = 'run_cells'
        elif plan.run_patches() is not None:
            cell = 'runs'
    if _long_rows and LONG_WAVE_FIELDS < n_batch * k_inner <= \
            LONG_WAVE_MAX and plan._wave is not None and LONG_WAVE_ROWS:
        cell = 'wave'
    args = plan._prefilled(whole, cell)
    if cell == 'wave':
        # many fields on long rows: one wave per row, the cells a few rows
        # share sliding through LDS (family 11, spmm_longwave.h)
        tune = [11]
        flags |= FLAG_TUNE_HINT
    elif cell == 'runs':
        tune = [5]
        flags |= FLAG_TUNE_HINT
    elif cell == 'run_cells':
        # 4 columns per chunk of the batch-at-a-time kernel; with 5 or 6
        # levels 6: the batch is ONE chunk (config 3's map, (80, n, 6):
        # 0.57 -> 0.47 ms, (96, n, 5): 0.70 -> 0.67)
        tune = list(_RUNS_TUNE) if _RUNS_TUNE else \
            [7, 6 if k_inner >= 5 else 4, 2]
        flags |= FLAG_TUNE_HINT
    elif cell:
        # 4 fields per lane and LDS image: the workgroup stays on its patch
        # over a run of chunks (spmm_patchtime), two images in LDS -- config
        # 3's map, Infinity-Cache-cold, 4 / 8 fields: (120, nCells) 0.141 /
        # 0.146 ms, (12, nCells) 16.4 / 24 us; (60, 3.7 M cells) 1.13 / 1.20
        tune = [7, 4]
        if _CELL_TUNE:
            tune = [7] + list(_CELL_TUNE)
        if _long_rows:
            K = n_batch * k_inner
            if K <= LONG_WAVE_FIELDS:
                # few fields: one wave per (long row, a few columns) -- the
                # row's loads run lanes-across-entries, only its sum is a
                # chain (family 9, spmm_longrow.h).  1 deg -> 0.5 deg with
                # pole caps, the long rows' launch: K = 1 15 -> 3 us
                tune = [9, _LONG_WAVE_TT or 0]
            else:
                # many fields: the 256 rows of a patch share their source
                # cells, staged once (family 7); a long row is ONE dependent
                # chain: few fields per lane keep its steps short and the
                # workgroups many (us per apply with 1 / 2 / 4 fields per
                # lane: K = 64 72 / 65 / 83, K = 512 423 / 352 / 320)
                tune = [7, _LONG_TT or (2 if K <= 128 else 4)]
        flags |= FLAG_TUNE_HINT
    args.row_begin = row_begin
    args.row_end = end
    args.X = X.data_ptr()
    args.x_dtype = x_dtype
    args.mode = mode
    args.x_row_stride = x_row_stride
    args.x_batch_stride = x_batch_stride
    args.Y = Y.data_ptr()
    args.y_row_stride = y_row_stride
    args.y_batch_stride = y_batch_stride
    args.n_batch = n_batch
    args.k_inner = k_inner
    args.frac_b = plan.frac_b.data_ptr() if mode == MODE_FRACB else None
    args.threshold = float(threshold)
    args.mask_out = mask_out.data_ptr() if mask_out is not None else None
    if gate is not None:
        if gate.dtype != torch.int32 or gate.device != plan.device:
            raise TypeError('gate must be an int32 tensor on the plan device')
        args.gate = gate.data_ptr()
        args.gate_value = int(gate_value)
    args.flags = flags
    args.x_src_fold = int(x_src_fold)
    args.x_outer_stride = int(x_outer_stride)
    if not tune:
        # the plan's preference (auto_schedule); the library falls back to
        # its own choice where the preferred family cannot serve the call
        tune = plan.default_tune
        if isinstance(tune, dict):
            tune = tune.get(mode)
        if tune:
            args.flags = flags | FLAG_TUNE_HINT
    if tune:
        for i, v in enumerate(tune):
            args.tune[i] = int(v)
    stream = ctypes.c_void_p(
        torch.cuda.current_stream(plan.device).cuda_stream)
    if torch.cuda.current_device() == plan.device.index:
        _check(lib.remap_apply_f64(ctypes.byref(args), stream),
               'remap_apply_f64')
    else:
        with torch.cuda.device(plan.device):
            _check(lib.remap_apply_f64(ctypes.byref(args), stream),
                   'remap_apply_f64')


def _prod(seq):
    out = 1
    for s in seq:
        out *= int(s)
    return out


def in_place_addressable(shape, remap_axes):
    """
    Can a field of this shape be addressed in place (strides instead of
    permute copies)?  The source axes must be adjacent -- then every layout
    has its kernel: a run of >= 8 contiguous fields behind the source axes
    goes to the lanes-across-K kernels, shorter runs in several batches
    ((Time, nCells), the reference's most common input) to the
    lanes-across-rows kernel (``spmm_rowcell``), K <= 32 to the
    lane-per-(row, k) kernel.
    """
    ndim = len(shape)
    axes = [int(a) % ndim for a in remap_axes]
    lead = min(axes)
    return axes == list(range(lead, lead + len(axes)))


def remap_tensor(plan, dst_grid_dims, field, remap_axes, mode, threshold=0.0,
                 want_mask=False, flags=0, tune=None, out=None, gate=None,
                 gate_value=0, mask_out=None):
    """
    Device-level ``_remap_numpy_array``: ``field`` is a device tensor whose
    axes ``remap_axes`` hold the source grid; returns the float64 tensor with
    those axes replaced by ``dst_grid_dims`` at ``min(remap_axes)``
    (``remap_numpy.py:280-295``), NaN where the reference masks, and the
    uint8 mask (1 = masked) when ``want_mask``.

    ``dst_grid_dims`` is in C order; for a row shard the leading destination
    dimension is replaced by the shard's row count (the result is the flat
    slab of rows ``[plan.row_offset, plan.row_offset + plan.n_b)``).
    """
    torch = _torch()
    if hasattr(plan, 'shards'):
        # a parallel.MultiDeviceRemap standing where the plan stands
        # (Remapper(..., devices=[...])): rows sharded over several GPUs
        if out is not None or gate is not None or mask_out is not None or \
                tune is not None:
            raise ValueError('out / gate / mask_out / tune address ONE '
                             "device's launch; not with a multi-device plan")
        return plan.remap_tensor(dst_grid_dims, field, remap_axes, mode,
                                 threshold=threshold, want_mask=want_mask,
                                 flags=flags)
    remap_axes = [int(a) % field.ndim for a in remap_axes]
    ndim = field.ndim
    extra_axes = [ax for ax in range(ndim) if ax not in remap_axes]
    n_src = _prod(field.shape[ax] for ax in remap_axes)
    if n_src != plan.n_a:
        raise ValueError(
            f'the remapped axes hold {n_src} source cells but the mapping '
            f'has n_a = {plan.n_a}')
    if field.dtype not in (torch.float64, torch.float32):
        # scipy upcasts everything else to float64 before the product
        field = field.to(torch.float64)

    lead = min(remap_axes)
    contiguous_block = remap_axes == list(range(lead, lead + len(remap_axes)))
    lead_shape = [int(s) for s in field.shape[:lead]]
    tail_shape = [int(field.shape[ax]) for ax in extra_axes if ax > lead]
    n_batch = _prod(lead_shape)
    k_inner = _prod(tail_shape)
    direct = in_place_addressable(field.shape, remap_axes)
    sharded = plan.n_b != plan.n_b_global
    # rows stay flat for a shard (its rows are no whole grid) and when the
    # caller names no destination grid
    dst_shape = [plan.n_b] if sharded or dst_grid_dims is None \
        else [int(d) for d in dst_grid_dims]
    if not sharded and _prod(dst_shape) != plan.n_b:
        raise ValueError(f'dst_grid_dims {dst_shape} do not hold n_b = '
                         f'{plan.n_b} cells')
    out_shape = lead_shape + dst_shape + tail_shape
    if out is not None:
        # the kernel writes through out.data_ptr(): anything but a float64,
        # contiguous tensor of exactly the result's shape on the plan's
        # device would be an out-of-bounds device write
        if out.dtype != torch.float64 or out.device != plan.device or \
                tuple(out.shape) != tuple(out_shape) or \
                not out.is_contiguous():
            raise ValueError(
                f'out must be a contiguous float64 tensor of shape '
                f'{tuple(out_shape)} on {plan.device}, got '
                f'{out.dtype} {tuple(out.shape)} on {out.device}')

    if mask_out is not None and (
            mask_out.dtype != torch.uint8 or mask_out.device != plan.device or
            tuple(mask_out.shape) != tuple(out_shape) or
            not mask_out.is_contiguous()):
        raise ValueError(
            f'mask_out must be a contiguous uint8 tensor of shape '
            f'{tuple(out_shape)} on {plan.device}')

    if direct:
        # strides do the permute/flatten of remap_numpy.py:254-256
        X = field.contiguous()
        Y = out if out is not None else torch.empty(
            out_shape, dtype=torch.float64, device=field.device)
        mask = None
        if want_mask:
            mask = mask_out if mask_out is not None else torch.empty(
                out_shape, dtype=torch.uint8, device=field.device)
        apply_strided(
            plan, X, Y, n_batch=n_batch, k_inner=k_inner,
            x_row_stride=k_inner, x_batch_stride=plan.n_a * k_inner,
            y_row_stride=k_inner, y_batch_stride=plan.n_b * k_inner,
            mode=mode, threshold=threshold, mask_out=mask, flags=flags,
            tune=tune, gate=gate, gate_value=gate_value)
        return (Y, mask) if want_mask else Y
    if len(remap_axes) == 2 and remap_axes[0] < remap_axes[1] and \
            out is None and gate is None and mask_out is None:
        # Two source axes with other dims BETWEEN them -- (lat, M, lon[, T])
        # -- in place too: the dims between are the batches of the launch,
        # a source cell is addressed through two strides (x_src_fold), and
        # the lanes-across-rows kernels do the rest; one launch per index of
        # the dims in front.  (remap_numpy.py:254-256 transposes.)
        a0, a1 = remap_axes
        X = field.contiguous()
        lead_n = _prod(X.shape[:a0])
        ny, nx = int(X.shape[a0]), int(X.shape[a1])
        M = _prod(X.shape[a0 + 1:a1])
        T = _prod(X.shape[a1 + 1:])
        between = [int(s) for s in X.shape[a0 + 1:a1]]
        trailing = [int(s) for s in X.shape[a1 + 1:]]
        full_shape = [int(s) for s in X.shape[:a0]] + dst_shape + \
            between + trailing
        Y = torch.empty(full_shape, dtype=torch.float64, device=X.device)
        mask = torch.empty(full_shape, dtype=torch.uint8,
                           device=X.device) if want_mask else None
        per_x, per_y = ny * M * nx * T, _prod(dst_shape) * M * T
        Xl = X.reshape(lead_n, per_x)
        Yl = Y.reshape(lead_n, per_y)
        Ml = mask.reshape(lead_n, per_y) if want_mask else None
        for li in range(lead_n):
            apply_strided(
                plan, Xl[li], Yl[li], n_batch=M, k_inner=T,
                x_row_stride=T, x_batch_stride=nx * T,
                y_row_stride=M * T, y_batch_stride=T, mode=mode,
                threshold=threshold,
                mask_out=Ml[li] if want_mask else None, flags=flags,
                tune=tune, x_src_fold=nx, x_outer_stride=M * nx * T)
        return (Y, mask) if want_mask else Y
    if gate is not None or mask_out is not None:
        raise ValueError('gated launches and caller-supplied masks need the '
                         'source axes adjacent (in-place addressing)')

    # general axis order (or a very short contiguous run): one device
    # transpose to (n_a, K), the kernel, one transpose back
    K = _prod(field.shape[ax] for ax in extra_axes)
    X = field.permute(remap_axes + extra_axes).reshape(n_src, K).contiguous()
    Y = torch.empty((plan.n_b, K), dtype=torch.float64, device=field.device)
    mask = torch.empty((plan.n_b, K), dtype=torch.uint8,
                       device=field.device) if want_mask else None
    apply_strided(plan, X, Y, n_batch=1, k_inner=K, x_row_stride=K,
                  x_batch_stride=0, y_row_stride=K, y_batch_stride=0,
                  mode=mode, threshold=threshold, mask_out=mask, flags=flags,
                  tune=tune)
    extra_shape = [int(field.shape[ax]) for ax in extra_axes]
    n_dst = len(dst_shape)
    tail = list(range(n_dst, n_dst + len(extra_shape)))
    unpermute = tail[:lead] + list(range(n_dst)) + tail[lead:]

    def back(t):
        t = t.reshape(dst_shape + extra_shape).permute(unpermute).contiguous()
        return t
    Y = back(Y)
    if out is not None:
        out.copy_(Y)
        Y = out
    return (Y, back(mask)) if want_mask else Y


def scan_nan(x, flag):
    """
    Asynchronously OR 1 into ``flag`` (int32 device tensor, zeroed by the
    caller) if the float32/float64 device tensor ``x`` holds a NaN: the
    device half of ``remap_numpy.py:201-204``.  A ``flag`` of two elements
    also receives the KIND of the missing values (``remap_scan_nan_kinds``):
    ``flag[1]`` = 0 (no NaN), 1 (NaNs in whole aligned runs: whole cells of
    an ``(n_a, K)`` field) or 3 (NaNs column by column).
    """
    torch = _torch()
    lib = load_library()
    if not x.is_contiguous():
        raise ValueError('scan_nan needs a contiguous tensor')
    dtype = {torch.float64: DTYPE_F64, torch.float32: DTYPE_F32}[x.dtype]
    fn, name = (lib.remap_scan_nan_kinds, 'remap_scan_nan_kinds') \
        if flag.numel() >= 2 else (lib.remap_scan_nan, 'remap_scan_nan')
    with torch.cuda.device(x.device):
        _check(fn(_ptr(x), dtype, x.numel(), _ptr(flag),
                  _stream_ptr(x.device)), name)


def scan_nan_layout(x, n_rows, n_batch, k_inner, kinds):
    """
    The NaN scan with the field's layout (``remap_scan_nan_layout``): ``x`` a
    contiguous device tensor holding ``n_batch`` batches of ``n_rows`` source
    cells of ``k_inner`` contiguous values -- ``(Time, nCells, nVertLevels)``
    -- and ``kinds`` four zeroed int32: any NaN; whole cells missing (1) or
    not (3); the same mask in every batch (1) or not (3); the masked form that
    suits (0 none, 1 ``FLAG_CELL_MASKS``, 2 ``FLAG_BATCH_MASKS``, 3 neither).
    """
    torch = _torch()
    lib = load_library()
    if not x.is_contiguous() or x.numel() != n_batch * n_rows * k_inner:
        raise ValueError('scan_nan_layout needs a contiguous (n_batch, '
                         'n_rows, k_inner) tensor')
    if kinds.numel() < 4 or kinds.dtype != torch.int32:
        raise ValueError('kinds: four int32')
    dtype = {torch.float64: DTYPE_F64, torch.float32: DTYPE_F32}[x.dtype]
    with torch.cuda.device(x.device):
        _check(lib.remap_scan_nan_layout(
            _ptr(x), dtype, n_rows, n_batch, k_inner, k_inner,
            n_rows * k_inner, _ptr(kinds), _stream_ptr(x.device)),
            'remap_scan_nan_layout')


def cell_mask_form(plan):
    """Does ``plan`` run the masked mode faster with FLAG_CELL_MASKS when
    whole cells are missing (8-row groups: entry-rich mappings)?"""
    groups = getattr(plan, 'groups', None)
    return bool(groups) and groups.get('rows') == 8


def remap_tensor_auto_mode(plan, dst_grid_dims, field, remap_axes, threshold,
                           flags=0, out=None, flag=None):
    """
    ``_remap_data_array``'s branch (``remap_numpy.py:201-204``) without a
    host round trip: the masked, renormalised result if ``field`` holds a
    NaN, the ``frac_b``-normalised one if not.  One scan, two gated launches
    (the one whose gate is closed does nothing; three on entry-rich mappings,
    whose masked branch has two forms), nothing synchronises.
    """
    torch = _torch()
    if hasattr(plan, 'shards'):
        if out is not None or flag is not None:
            raise ValueError("out / flag address ONE device's launch; not "
                             'with a multi-device plan')
        return plan.remap_tensor_auto_mode(dst_grid_dims, field, remap_axes,
                                           threshold, flags=flags)
    if not in_place_addressable(field.shape, remap_axes):
        # permute copies either side of the launch: decide with one readback
        masked = bool(torch.isnan(field).any())
        return remap_tensor(plan, dst_grid_dims, field, remap_axes,
                            MODE_MASKED if masked else MODE_FRACB,
                            threshold=threshold if masked else 0.0,
                            flags=flags, out=out)
    X = field.contiguous()
    if flag is None and cell_mask_form(plan):
        # entry-rich mapping: the scan -- told where the cells and the
        # batches of the field are -- also says whether whole cells are
        # missing (land) or the mask is the same in every batch
        # (bathymetry), and the masked branch comes in the form that suits
        # (REMAP_FLAG_CELL_MASKS / REMAP_FLAG_BATCH_MASKS / neither): up to
        # four gated launches, one of which runs
        axes = [int(a) % X.ndim for a in remap_axes]
        lead = min(axes)
        n_batch = _prod(X.shape[:lead])
        k_inner = _prod(X.shape[lead + len(axes):])
        hints = FLAG_CELL_MASKS | FLAG_BATCH_MASKS
        flags &= ~hints
        kinds = torch.zeros(4, dtype=torch.int32, device=field.device)
        scan_nan_layout(X, plan.n_a, n_batch, k_inner, kinds)
        Y = remap_tensor(plan, dst_grid_dims, X, remap_axes, MODE_MASKED,
                         threshold=threshold, flags=flags | FLAG_CELL_MASKS,
                         out=out, gate=kinds[3:], gate_value=1)
        if n_batch >= 3:
            Y = remap_tensor(plan, dst_grid_dims, X, remap_axes, MODE_MASKED,
                             threshold=threshold,
                             flags=flags | FLAG_BATCH_MASKS, out=Y,
                             gate=kinds[3:], gate_value=2)
        Y = remap_tensor(plan, dst_grid_dims, X, remap_axes, MODE_MASKED,
                         threshold=threshold, flags=flags, out=Y,
                         gate=kinds[3:], gate_value=3)
        return remap_tensor(plan, dst_grid_dims, X, remap_axes, MODE_FRACB,
                            flags=flags, out=Y, gate=kinds, gate_value=0)
    if flag is None:
        flag = torch.zeros(1, dtype=torch.int32, device=field.device)
    scan_nan(X, flag)
    Y = remap_tensor(plan, dst_grid_dims, X, remap_axes, MODE_MASKED,
                     threshold=threshold, flags=flags, out=out, gate=flag,
                     gate_value=1)
    return remap_tensor(plan, dst_grid_dims, X, remap_axes, MODE_FRACB,
                        flags=flags, out=Y, gate=flag, gate_value=0)


def gather_rows(field, axis, rows, out=None):
    """
    ``field.index_select(axis, rows)`` for a contiguous device tensor through
    the library's ``remap_gather_rows``: the packed source buffer
    ``X[ucols]`` a row shard's GPU receives.  ``rows``: int32 device tensor.
    Asynchronous on torch's current stream.
    """
    torch = _torch()
    lib = load_library()
    if not field.is_contiguous():
        field = field.contiguous()
    axis = int(axis) % field.ndim
    if rows.dtype != torch.int32 or rows.device != field.device:
        rows = rows.to(device=field.device, dtype=torch.int32)
    rows = rows.contiguous()
    n_batch = _prod(field.shape[:axis])
    inner = _prod(field.shape[axis + 1:])
    item = field.element_size()
    shape = list(field.shape[:axis]) + [int(rows.shape[0])] + \
        list(field.shape[axis + 1:])
    if out is None:
        out = torch.empty(shape, dtype=field.dtype, device=field.device)
    elif list(out.shape) != shape or out.dtype != field.dtype or \
            out.device != field.device or not out.is_contiguous():
        raise ValueError(f'out must be a contiguous {field.dtype} tensor of '
                         f'shape {tuple(shape)} on {field.device}')
    with torch.cuda.device(field.device):
        _check(lib.remap_gather_rows(
            _ptr(field), n_batch, int(field.shape[axis]) * inner * item,
            inner * item, _ptr(rows), int(rows.shape[0]), inner * item,
            _ptr(out), _stream_ptr(field.device)), 'remap_gather_rows')
    return out


def clock_probe(device, micros=20):
    """
    Enqueue a shader-clock measurement on torch's current stream of
    ``device`` (``remap_clock_probe``: one wave spinning for ``micros`` us).
    Returns a function that, once the stream has been synchronised, gives
    the clock in MHz the chip held at that point of the stream.
    """
    torch = _torch()
    lib = load_library()
    device = torch.device(device)
    ticks = torch.zeros(2, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _check(lib.remap_clock_probe(_ptr(ticks), int(micros),
                                     _stream_ptr(device)),
               'remap_clock_probe')

    def mhz():
        t, r = (int(v) for v in ticks.cpu())
        return 100.0 * t / r if r else float('nan')
    return mhz


def stream_copy(dst, src):
    """Asynchronous device copy through the library's streaming kernel."""
    torch = _torch()
    lib = load_library()
    nbytes = src.numel() * src.element_size()
    with torch.cuda.device(src.device):
        _check(lib.remap_stream_copy(_ptr(dst), _ptr(src), nbytes,
                                     _stream_ptr(src.device)),
               'remap_stream_copy')
