/*
 * remap_hip.h -- C ABI of libremap_hip.so, the MI355X (gfx950) weight-
 * application engine behind pyremap's Remapper.remap_numpy()/ncremap().
 *
 * The reference (MPAS-Dev/pyremap v2.4.0) is pure Python and has no FFI
 * boundary of its own; the seam this library replaces is the arithmetic of
 *
 *   pyremap/remapper/remap_numpy.py:134-137   COO triplets -> scipy CSR
 *                                             -> remap_csr_from_coo()
 *   pyremap/remapper/remap_numpy.py:258-278   masked / unmasked SpMM,
 *                                             normalisation and masking
 *                                             -> remap_apply_f64()
 *   pyremap/remapper/remap_numpy.py:254-256,  permute/flatten and
 *                                   280-295   unflatten/unpermute: absorbed
 *                                             into the batch/row strides of
 *                                             remap_apply_args (no copies)
 *   pyremap/remapper/remap_numpy.py:201-204   `isnan(values).any()` picks the
 *                                             masked or the unmasked branch
 *                                             -> remap_scan_nan() + two calls
 *                                             gated on its flag
 *                                             (remap_apply_args.gate)
 *   (no counterpart in the reference)         which kernel schedule a mapping
 *                                             gets, and building it on the
 *                                             device -> remap_schedule_auto()
 *                                             (remap_groups_build() and
 *                                             remap_patches_build() are its
 *                                             two builders, also exported)
 *   pyremap/remapper/remap_numpy.py:72-139    `_load_mapping` as a whole and
 *                                   223-297   `_remap_numpy_array` as a whole
 *                                             behind one opaque handle whose
 *                                             device memory the library owns
 *                                             -> remap_plan_create() /
 *                                             remap_plan_apply() /
 *                                             remap_plan_destroy()
 *
 * Conventions: extern "C"; plain pointers and sizes only; every pointer
 * marked (device) is an address in the current HIP device's memory (e.g.
 * torch.Tensor.data_ptr()); all launches are asynchronous on the caller's
 * hipStream_t (passed as void*; NULL = the null stream); nothing is allocated,
 * freed or synchronised inside the apply calls, so they are graph-capturable
 * (only remap_plan_create / _destroy allocate and free device memory).
 * Return value: REMAP_OK (0) or a negative REMAP_ERR_*; the message of the
 * last failure on the calling thread is available from remap_last_error().
 *
 * The reference-side binding (ctypes) is in pyremap_amd/engine.py; the stub a
 * pyremap maintainer would add is shown in INTEGRATION.md.
 */
#ifndef REMAP_HIP_H
#define REMAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REMAP_ABI_VERSION 25

/* The library is built with -fvisibility=hidden: the entry points declared
 * here, and nothing else, are its dynamic symbols. */
#if defined(__GNUC__) || defined(__clang__)
#define REMAP_API __attribute__((visibility("default")))
#else
#define REMAP_API
#endif

enum {
    REMAP_OK = 0,
    REMAP_ERR_ARG = -1,         /* bad argument (NULL, negative size, ...)  */
    REMAP_ERR_UNSUPPORTED = -2, /* valid request this build cannot serve    */
    REMAP_ERR_HIP = -3,         /* a HIP runtime call or launch failed      */
    REMAP_ERR_WORKSPACE = -4    /* workspace too small                      */
};

/* element type of the source field X (the output is always float64, as in the
 * reference: S is float64 and scipy upcasts, remap_numpy.py:264-268) */
enum { REMAP_DTYPE_F64 = 0, REMAP_DTYPE_F32 = 1 };

/* what is computed after num = A.X */
enum {
    /* Y = A.X, nothing else (the bare `matrix.dot`, remap_numpy.py:268) */
    REMAP_MODE_RAW = 0,
    /* unmasked branch, remap_numpy.py:268-278: den = frac_b[row];
     * Y = den > 0 ? num / den : NaN.  NaNs in X propagate. */
    REMAP_MODE_FRACB = 1,
    /* masked branch, remap_numpy.py:262-266,277-278, with the mask taken
     * from NaN in X (as _remap_data_array builds it, :201-204):
     * num = A.(X, NaN->+0), den = A.(!isnan X); Y = den > threshold ?
     * num / den : NaN. */
    REMAP_MODE_MASKED = 2
};

enum {
    /* accumulate with fused multiply-add.  Default (flag clear) is a separate
     * multiply and add in CSR order, which is bit-identical to scipy's
     * csr_matvecs; FMA is within 1 ulp per term of it, not identical. */
    REMAP_FLAG_FMA = 1u << 0,
    /* bit 1 is unused (it once asked for write-back instead of non-temporal
     * stores of Y; non-temporal measured faster on every configuration) */
    /* `tune` is a preference, not a demand: if the requested kernel family
     * cannot serve this call (K <= 32, odd strides, a partial row range, a
     * missing schedule ...) choose automatically instead of failing */
    REMAP_FLAG_TUNE_HINT = 1u << 2,
    /* calls served by kernel family 3 (K <= 32, opt-in) may add a row's products
     * by lane-private partial sums and a butterfly across lanes instead of
     * one after the other in CSR order: a different association, within
     * 1e-13 relative of the default -- for callers that do not need the
     * bits.  Ignored by the other kernel families. */
    REMAP_FLAG_TREE = 1u << 3,
    /* REMAP_MODE_MASKED, a HINT: the caller expects a source cell to be
     * valid in all of the call's columns or missing in all of them (land, an
     * ice shelf, no NaN at all in most of the field) rather than column by
     * column (a 3-D field cut by bathymetry).  On mappings scheduled as
     * 8-row groups (entry-rich: 2nd-order conservative) the normaliser
     * `A . [not isnan X]` (remap_numpy.py:265) is then kept per ROW, with
     * two K tiles per wave as in the frac_b mode; groups where the
     * expectation fails are redone with per-lane normalisers inside the
     * same launch.  Same bits with or without the flag, whatever the data;
     * only the speed differs (config 5: 27.4 -> 22.0 ms with whole cells
     * missing, 26.7 -> 35.3 with the flag set on a bathymetry mask).
     * remap_scan_nan_kinds() tells the two apart on the device.  Ignored
     * elsewhere. */
    REMAP_FLAG_CELL_MASKS = 1u << 4,
    /* REMAP_MODE_MASKED on a field of several batches, a HINT (ABI 25): the
     * mask is expected to be the same in every batch -- a (Time, nCells,
     * nVertLevels) ocean variable is missing below the sea floor at every
     * time.  The reference sums the normaliser `A . [not isnan X]` for every
     * column of every time slice (remap_numpy.py:265); on mappings scheduled
     * as 8-row groups a wave then takes four time slices of a level per lane
     * and keeps ONE normaliser for them (csrc/spmm_grouptime.h).  Groups
     * where the expectation fails are redone with per-element normalisers
     * inside the same launch: same bits with or without the flag, whatever
     * the data.  Needs n_batch >= 3; ignored elsewhere.  The device-side
     * scan remap_scan_nan_layout() tells whether it holds. */
    REMAP_FLAG_BATCH_MASKS = 1u << 5
};

/* CSR weight matrix of shape (n_rows, n_cols) = (n_b, n_a), or a row shard of
 * it (rows [r0, r1) of the full matrix with rowptr rebased to 0). */
typedef struct remap_csr {
    int64_t n_rows;
    int64_t n_cols;
    int64_t nnz;
    const int64_t *rowptr; /* (device) n_rows + 1                            */
    const int32_t *col;    /* (device) nnz, ascending within a row, 0-based  */
    const double *val;     /* (device) nnz                                   */
    int64_t max_row_nnz;   /* entries of the longest row, 0 = unknown (lets
                              the library pick kernels specialised for short
                              rows; a wrong value only costs speed if too
                              large, but MUST NOT be smaller than the truth) */
    int64_t csr_pad;       /* readable (don't-care) entries allocated behind
                              col[nnz-1] and val[nnz-1]; >= 8 enables the
                              kernels that fetch a row's entries 8 at a time
                              through the scalar cache                       */
} remap_csr;

/*
 * Optional strip schedule of kernel family 8 (remap_apply_args.strips; built
 * by the host layer: pyremap_amd.engine.RemapPlan.build_strips).  For
 * entry-rich mappings onto a 2-D destination grid (2nd-order conservative
 * stencils: 12-30 entries per row): the grid is cut into strips of
 * `strip_rows` grid rows, a strip into segments, a segment is walked in steps
 * of `step_cols` grid columns.  Unit u = one (strip, segment); its steps are
 * g = u * steps_per_unit + t, t < unit_steps[u].  A workgroup owns one (unit,
 * 64-column K-chunk): the 512-byte pieces of the source rows a step needs
 * ARRIVE in an LDS ring of `ring_slots` slots, `depth` steps ahead, and stay
 * while consecutive steps use them; the rows of a step are dealt to 8
 * compute waves, `rows_per_wave` each.  Every row still adds its entries in
 * ascending column order: results are unchanged.
 * The schedule is TRUSTED: the library checks the scalar fields only.  At
 * most 64 arrival pairs per step, arrival slots below ring_slots / 2, record
 * offsets and meta blocks inside meta_slot_bytes are the builder's to keep
 * (pyremap_amd/strips.py does, tests/test_strips_cpu.py replays it); a
 * schedule that breaks them makes the LDS-DMA write outside the ring.  Build
 * it with pyremap_amd.strips.build_strips or leave `strips` NULL.
 */
typedef struct remap_strips {
    int64_t n_units;
    int32_t steps_per_unit;
    int32_t rows_per_wave;      /* ceil(strip_rows * step_cols / waves)      */
    int32_t ring_slots;         /* piece slots, even: ring_slots * 512 bytes */
    int32_t depth;              /* steps an arrival is issued ahead: 1 .. 6
                                   (= loader waves per workgroup)            */
    int32_t meta_slot_bytes;    /* largest meta block, rounded up to 1 KiB;
                                   (depth + 1) such slots follow the piece
                                   slots and 1 KiB holding a slot of zeros   */
    int32_t waves;              /* compute waves per workgroup the rows of a
                                   step are dealt to; waves + depth <= 16    */
    const int32_t *unit_steps;  /* (device) n_units                          */
    /* arrival pairs of step g: i in [arr_ptr[g], arr_ptr[g + 1]); pair i
     * brings the pieces of source rows arr_src[2 i] and arr_src[2 i + 1]
     * (-1: none) into slots 2 arr_slot[i] and 2 arr_slot[i] + 1. */
    const int32_t *arr_ptr;     /* (device) n_units * steps_per_unit + 1     */
    const int32_t *arr_src;     /* (device) 2 per pair                       */
    const int32_t *arr_slot;    /* (device) 1 per pair                       */
    /* meta block of step g: 16-byte units [meta_ptr[g], meta_ptr[g + 1]) of
     * `meta`.  First waves * rows_per_wave row headers of 32 bytes, work
     * slot wave * rows_per_wave + i: {int32 row id (-1: none), int32 first
     * record, int32 records (a multiple of 4), 0; double frac_b, 0}; then
     * the step's records, 16 bytes each, a row's entries in ascending column
     * order followed by its pads: {int32 byte offset of the source row's
     * piece in LDS (slot * 512; pads: ring_slots * 512, the slot of zeros),
     * 0; double weight (pads: +0.0)}. */
    const int64_t *meta_ptr;    /* (device) n_units * steps_per_unit + 1     */
    const void *meta;           /* (device) 16-byte aligned (+ 1 KiB readable) */
} remap_strips;

/*
 * Two levels.  The PUBLIC, frozen face for a binder is
 * the opaque plan handle further down -- remap_plan_create / _apply /
 * _destroy with the 15-field remap_field -- which hides every schedule
 * internal.  remap_apply_args below is the ADVANCED level: the non-allocating
 * entry points a host layer with its own allocator composes (this package's
 * Python layer does); its schedule fields mirror remap_schedule one to one
 * and are filled by remap_schedule_auto, not by hand.
 *
 * One application of the weights to a batch of fields.
 *
 * The flattened (n_a, K) matrix of the reference (remap_numpy.py:254-256) is
 * addressed in place: K = n_batch * k_inner and flat column kf = b * k_inner
 * + k lives at X[b * x_batch_stride + a * x_row_stride + k] (strides in
 * elements, the k index contiguous).  A field laid out (n_a, K) has
 * n_batch = 1; an MPAS field (Time, nCells, nVertLevels) has n_batch = Time,
 * k_inner = nVertLevels, x_row_stride = nVertLevels, x_batch_stride =
 * nCells * nVertLevels.  Y is addressed the same way with its own strides, so
 * the output lands directly in the reference's final axis order
 * (remap_numpy.py:280-295).
 */
typedef struct remap_apply_args {
    remap_csr A;
    int64_t row_begin;      /* rows [row_begin, row_end) of A are computed;  */
    int64_t row_end;        /* Y / frac_b / mask_out are indexed by row      */
    const void *X;          /* (device) source field                         */
    int32_t x_dtype;        /* REMAP_DTYPE_*                                 */
    int32_t mode;           /* REMAP_MODE_*                                  */
    int64_t x_row_stride;
    int64_t x_batch_stride;
    double *Y;              /* (device) destination field, float64           */
    int64_t y_row_stride;
    int64_t y_batch_stride;
    int64_t n_batch;
    int64_t k_inner;
    const double *frac_b;   /* (device) n_rows; REMAP_MODE_FRACB only        */
    double threshold;       /* REMAP_MODE_MASKED only                        */
    uint8_t *mask_out;      /* (device) optional, addressed like Y; 1 where
                               the reference's result is masked (~ok)        */
    const int32_t *row_order; /* (device) optional processing order: entry s
                               (row_begin <= s < row_end) names the row that
                               work slot s computes; must be a permutation of
                               [row_begin, row_end).  Scheduling only -- the
                               results do not depend on it.  Used to walk a
                               2-D destination grid in tiles so neighbouring
                               rows (which share source rows) run together
                               and re-touches hit the XCD's L2.             */
    /* Optional LDS-staging schedule ("patch plan", all NULL/0 = absent).
     * Work slots [row_begin + j*patch_rows, +patch_rows) form patch j (with
     * row_order, a 2-D tile of the destination grid).
     *   patch_ucol   per patch, the DISTINCT source rows its entries
     *                reference (ascending); patch_ptr[j] .. patch_ptr[j+1]
     *                delimits patch j's list;
     *   patch_rowptr / patch_lidx / patch_val: the weights again, laid out
     *                in SLOT order (patch-major CSR): slot s owns entries
     *                patch_rowptr[s - row_begin] .. patch_rowptr[s - row_begin
     *                + 1], in the row's CSR order; patch_lidx[e] is the
     *                position of that entry's column in its patch's list and
     *                patch_val[e] its weight.
     * The kernel fetches each distinct source row ONCE per patch into LDS
     * (together with the patch's entries) and serves the patch's
     * ~nnz/n_a-fold re-touches from there.  Results are unaffected (same
     * per-row summation order).                                          */
    const int32_t *patch_ptr;    /* (device) n_patches + 1                  */
    const int32_t *patch_ucol;   /* (device) patch_ptr[n_patches]           */
    const int32_t *patch_rowptr; /* (device) row_end - row_begin + 1        */
    const int32_t *patch_lidx;   /* (device) nnz                            */
    const double *patch_val;     /* (device) nnz                            */
    int32_t patch_rows;          /* work slots per patch                    */
    int32_t patch_umax;          /* longest per-patch list (sizes the LDS)  */
    int32_t patch_emax;          /* most entries in one patch               */
    int32_t patch_row_bytes;     /* staged bytes per source row and K-chunk:
                                    1024 (128 columns) or 512 (64 columns)   */
    int64_t n_patches;
    /* Optional row-group schedule (all NULL/0 = absent): with G =
     * group_rows (8 or 4), work slots [row_begin + G g, + G) form group g
     * (with row_order: a 2 x 4 or 2 x 2 tile of the destination grid).
     * group_col lists, per group, the sorted UNION of its rows' columns;
     * bit m of group_mask[u] says whether the group's m-th row owns union
     * entry u; group_w holds the weights of exactly those present (entry,
     * member) pairs, in (entry, member) order -- nnz doubles in all;
     * group_meta[2 g] / [2 g + 1] index group g's first union entry / first
     * weight (entry n_groups closes the lists).  group_rid / group_frac give
     * the row id / frac_b of every work slot, padded to whole groups (the
     * pad names any valid row).  A wave then loads every distinct source
     * row of G neighbouring destination rows ONCE and feeds up to G
     * accumulators from it; each row still adds its entries in ascending
     * column order, so results are unchanged.                              */
    const int64_t *group_meta;  /* (device) 2 * (n_groups + 1)              */
    const int32_t *group_col;   /* (device) union entries (+ 32 readable)   */
    const double *group_w;      /* (device) present weights (+ 128 readable) */
    const int32_t *group_mask;  /* (device) union entries (+ 32)            */
    const int32_t *group_rid;   /* (device) n_groups * group_rows           */
    const double *group_frac;   /* (device) n_groups * group_rows           */
    int64_t n_groups;
    int32_t group_rows;         /* G: 8 or 4 (16: float64, even strides)    */
    int32_t group_reserved;     /* must be 0                                */
    /* Optional device-side switch (NULL = always run): the launch does its
     * work only if the int32 at `gate` equals `gate_value` when the kernel
     * starts, and is a no-op otherwise.  With remap_scan_nan() this moves
     * the reference's `isnan(values).any()` decision (remap_numpy.py:201-204:
     * masked branch iff the field holds a NaN) onto the device: scan into a
     * flag, then enqueue the MASKED call gated on 1 and the FRACB call gated
     * on 0 -- no host synchronisation in between.                           */
    const int32_t *gate;    /* (device) optional                             */
    int32_t gate_value;
    uint32_t flags;         /* REMAP_FLAG_*                                  */
    /* launch tuning, 0 = choose automatically:
     * tune[0] kernel family   1 = wave per row (lanes across K),
     *                         2 = lane per (row, k) (small K),
     *                         3 = a sub-group of lanes per row, lanes across
     *                             the row's entries (K <= 32, opt-in: it
     *                             measured slower than 2; tune[1] = lanes
     *                             per row, 8 or 4),
     *                         5 = LDS-staged patches (needs a patch plan),
     *                         10 = G rows per wave over the union of their
     *                              columns (needs the row-group schedule),
     *                         6 = 1 with row metadata through the scalar
     *                             cache (needs csr_pad >= 8; the default)
     *                         4 = lanes across destination rows, TT fields
     *                             per lane (short contiguous runs in
     *                             several batches: (Time, nCells)),
     *                         7 = 4 with the distinct source cells of a
     *                             patch staged in LDS (needs a patch plan)
     *                         8 = an LDS ring sliding along strips of the
     *                             destination grid (needs `strips`)
     *                         9 = one wave per (row, few columns), the
     *                             row's entries read lanes-across-entries
     *                             and summed in order from LDS: for a CSR of
     *                             LONG rows (hundreds of entries) and few
     *                             fields.  `A` then holds those rows only:
     *                             its row r is work slot r and row_order[r],
     *                             if given, names the row of Y / frac_b /
     *                             mask_out it writes; needs A.max_row_nnz;
     *                             tune[1] = columns per wave (1 ... 16)
     *                         11 = one wave per LONG row x 64 columns, the
     *                             distinct source cells of a patch of R <=
     *                             16 consecutive long rows sliding through
     *                             LDS in windows of 8 R cells (many fields;
     *                             `A` and row_order as for 9).  Needs a
     *                             patch plan (remap_patches_build, tile 1 x
     *                             R, no patch_ell_base) whose patch_lidx do
     *                             not decrease inside a row -- true when the
     *                             rows of A are sorted by column
     * tune[1] doubles per lane per tile (1 or 2); family 10: waves per
     *         workgroup (1, 2; else 4); family 2: entries of a row fetched
     *         together (1, 4 or 8); family 5: threads per workgroup on
     *         64-column chunks (512 or 1024; 0: by the length of the work
     *         list)
     * tune[2] K tiles per wave (1, 2 or 4)
     * tune[3] consecutive rows per wave
     * tune[4] block -> work map: 1 = as dispatched, 2 = XCD-contiguous;
     *         family 10 also 3 = XCD-contiguous with the K-chunks of a row
     *         block side by side in the work list (the schedule of a group
     *         is fetched once per XCD: what remap_schedule_auto picks for
     *         entry-rich mappings)
     * tune[5] family 10: union entries in flight per wave (8; or 4, 16);
     *         26 / 28: the rolling form with 6 / 8 in flight (float64, even
     *         strides; spmm_grouproll.h: measured, not chosen); 9: keep the
     *         per-lane masked form under REMAP_FLAG_CELL_MASKS; 8 (the
     *         default count, said aloud): keep the 8-row groups under
     *         REMAP_FLAG_CELL_MASKS on a plan with share_* lists; 32: the
     *         shared form (share_* below: the frac_b and raw modes;
     *         tune[2] = K tiles per wave, 1 or 2)
     * tune[6..7] reserved, must be 0 (a -DREMAP_DIAG build of the library,
     *         tools/build_diag.py, reads bottleneck-analysis switches from
     *         them; this build rejects them) */
    int32_t tune[8];
    /* Two source axes that are NOT adjacent in memory -- a field (lat, M,
     * lon[, T]) remapped over (lat, lon): source cell a = y * x_src_fold + x
     * lives at y * x_outer_stride + x * x_row_stride (+ the column's offset
     * b * x_batch_stride + k, where the dims BETWEEN the two axes are the
     * batches: n_batch = M, x_batch_stride = their stride, k_inner = T).
     * x_src_fold = 0 (the default): one stride, a * x_row_stride.  Served in
     * place by the lanes-across-rows kernels (families 4 and 7); the
     * reference takes a transpose copy (remap_numpy.py:254-256). */
    int64_t x_src_fold;
    int64_t x_outer_stride;
    /* Family 7 only, optional (NULL: the patch plan's entries lie row after
     * row, patch_rowptr addressing them).  Not NULL: they lie COLUMN-MAJOR
     * inside every patch -- entry j of the patch's slot r at
     * patch_val / patch_lidx [patch_ell_base[patch] + j * patch_rows + r];
     * patch_rowptr then only gives the slots' entry counts -- so that the 64
     * lanes of a wave, one slot each, read their j-th entries with ONE
     * coalesced load.  For patches of LONG rows: the pole caps of a global
     * bilinear map as ESMF makes it, 362-1 442 entries per row, where a
     * lane reading its own row's entries one by one pays a line fetch per
     * lane and entry (engine.RemapPlan._split_long_rows). */
    const int64_t *patch_ell_base;
    /* Optional strip schedule (HOST pointer to a struct of device pointers;
     * NULL = absent), kernel family 8: the whole row range, float64 X, one
     * batch (n_batch == 1) of an even number of contiguous columns. */
    const remap_strips *strips;
    /* Optional SHARED union lists on top of an 8-row group schedule (all
     * NULL/0 = absent; ABI 25): share_waves (2 or 4) consecutive groups -- a
     * 4 x 4 or 4 x 8 tile of the destination grid when the groups were built
     * with super_tile = 4 or 8 -- form a SUPERGROUP served by one workgroup
     * of share_waves waves.  share_col lists, per supergroup, the sorted
     * union of the source rows its 8 * share_waves rows reference; bit
     * 8 w + m of share_mask[u] says whether member m of the supergroup's
     * w-th group owns union entry u; share_meta[2 s] is supergroup s's first
     * union entry (entry n_super closes the lists; the odd slots are not
     * read).  The weights are group_w of the 8-row schedule: its
     * (group, entry, member) order is every wave's own contiguous stream.
     * The workgroup sends each distinct source row of the supergroup ONCE
     * from global memory into an LDS ring (LDS-DMA) and every wave adds the
     * entries its own rows own from there (csrc/spmm_groupshare.h); a row
     * still adds its entries in ascending column order, so results are
     * unchanged.  Used by family 10 on float64 fields with even strides:
     * with tune[5] = 32 in the frac_b and raw modes on at least 104
     * columns (one K tile per wave up to 128, two beyond) and on 34 ... 64
     * columns (a lane per column: csrc/spmm_narrowshare.h), with
     * REMAP_FLAG_BATCH_MASKS (csrc/spmm_timeshare.h) or, on
     * more than 128 columns, REMAP_FLAG_CELL_MASKS (csrc/spmm_cellshare.h)
     * in the masked mode (share_waves = 4: the shape the kernels are built
     * in; remap_share_build also makes the lists of 2 groups).             */
    const int64_t *share_meta;  /* (device) 2 * (n_super + 1)               */
    const int32_t *share_col;   /* (device) union entries (+ 256 readable)  */
    const int32_t *share_mask;  /* (device) union entries (+ 256 readable)  */
    int32_t share_waves;        /* 2 or 4                                   */
    int32_t share_reserved;     /* must be 0                                */
} remap_apply_args;

/* ABI / build information */
REMAP_API int remap_abi_version(void);
REMAP_API const char *remap_arch(void);       /* "gfx950" */
REMAP_API const char *remap_last_error(void); /* thread-local, never NULL */

/* Number of visible HIP devices, or a negative REMAP_ERR_HIP. */
REMAP_API int remap_device_count(void);

/*
 * Apply the weights (asynchronous on `stream`).  Replaces
 * remap_numpy.py:258-278 (and, through the strides, :254-256 and :280-295).
 */
REMAP_API int remap_apply_f64(const remap_apply_args *args, void *stream);

/*
 * COO -> CSR on the device: scipy's `csr_matrix((S, (row, col)))` of
 * remap_numpy.py:134-137.  Stable sort by (row, col), duplicates summed in
 * input order, explicit zeros kept.  `index_base` is subtracted from row and
 * col (1 for a SCRIP/ESMF mapping file).
 *
 *   rowptr_out (device) n_rows + 1;  col_out / val_out (device) nnz entries
 *   (only the first *nnz_out are meaningful);  nnz_out (device) one int64;
 *   bad_out (device) one int64: number of triplets whose row or col is out
 *   of range (the caller must treat non-zero as an error).
 *
 * All outputs are written asynchronously on `stream`.  Query the workspace
 * size first; the workspace is plain device memory owned by the caller.
 */
REMAP_API int remap_csr_from_coo_workspace(int64_t nnz, int64_t n_rows,
                                 size_t *bytes_out);
REMAP_API int remap_csr_from_coo(int64_t n_rows, int64_t n_cols, int64_t nnz,
                       const int32_t *row, const int32_t *col,
                       const double *S, int32_t index_base,
                       int64_t *rowptr_out, int32_t *col_out,
                       double *val_out, int64_t *nnz_out, int64_t *bad_out,
                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * Build the row-group schedule of kernel family 10 (remap_apply_args.group_*)
 * for the rows of A on the device -- what a binder needs to reach the
 * kernels that carry the headline numbers without any host-side logic of its
 * own.  Asynchronous on `stream`; nothing is allocated.
 *
 *   A            the CSR whose rows are scheduled (a row shard: its rows,
 *                rowptr rebased to 0); canonical (remap_csr_from_coo's form)
 *   frac_b       (device) A.n_rows
 *   group_rows   G: 8, or 4
 *   grid_dims    (HOST) NULL: groups are G consecutive rows; else {my, mx},
 *                the C-order dims of the WHOLE destination grid: groups are
 *                2 x G/2 tiles of it, walked row-major inside super_tile x
 *                super_tile blocks (super_tile <= 0: over the whole grid)
 *   row_offset   index of A's first row in the whole grid (0 unless a shard)
 *   share_waves  0, or (ABI 25, group_rows = 8) 2 / 4: the group tiles are
 *                walked inside 4 x 4 / 4 x 8 tiles -- 2 / 4 consecutive
 *                groups, the supergroups of remap_share_build -- and those
 *                row-major inside the supertiles (a multiple of the tile)
 *   row_order_out (device, A.n_rows, may be NULL when grid_dims is NULL) the
 *                processing order the schedule assumes: pass it as
 *                remap_apply_args.row_order together with the schedule
 *   group_meta   (device) 2 * (n_groups + 1)     n_groups = ceil(n_rows / G)
 *   group_col, group_mask  (device) A.nnz + 32 each (union entries <= nnz)
 *   group_w      (device) A.nnz + 128
 *   group_rid, group_frac  (device) n_groups * G
 *   n_union_out  (device) one int64: union entries actually used
 */
REMAP_API
int remap_groups_workspace(int64_t n_rows, int64_t nnz, size_t *bytes_out);
REMAP_API int remap_groups_build(const remap_csr *A, const double *frac_b,
                       int32_t group_rows, const int64_t *grid_dims,
                       int64_t row_offset, int32_t super_tile,
                       int32_t share_waves,
                       int32_t *row_order_out, int64_t *group_meta,
                       int32_t *group_col, int32_t *group_mask,
                       double *group_w, int32_t *group_rid,
                       double *group_frac, int64_t *n_union_out,
                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * Build the SHARED union lists of the shared form of family 10
 * (remap_apply_args.share_*; csrc/spmm_groupshare.h) on top of an 8-row
 * group schedule remap_groups_build has made for the same rows: supergroup s
 * = work slots [8 * share_waves * s, + 8 * share_waves) of that schedule
 * (build it with the same share_waves, so that a supergroup is a 4 x 4 /
 * 4 x 8 tile of the destination grid).  Asynchronous
 * on `stream`; nothing is allocated; the workspace is the one
 * remap_groups_workspace(A.n_rows, A.nnz) sizes.
 *
 *   group_rid    (device) the 8-row schedule's row of every work slot
 *   share_meta   (device) 2 * (n_super + 1),  n_super = ceil(n_rows / (8 *
 *                share_waves)); slot 2 s = first union entry of supergroup s
 *   share_col, share_mask  (device) A.nnz + 256 each (union entries <= nnz;
 *                the 256 behind the last one are readable zeros)
 *   n_union_out  (device) one int64: union entries actually used
 */
REMAP_API int remap_share_build(const remap_csr *A, const int32_t *group_rid,
                      int32_t share_waves, int64_t *share_meta,
                      int32_t *share_col, int32_t *share_mask,
                      int64_t *n_union_out, void *workspace,
                      size_t workspace_bytes, void *stream);

/*
 * Build the LDS patch plan of kernel family 5 (remap_apply_args.patch_*) for
 * the rows of A on the device.  Work slots are walked in tile_y x tile_x
 * tiles of a 2-D destination grid (grid_dims = HOST {my, mx}; NULL: natural
 * order, patches of tile_y * tile_x consecutive rows); tile_y * tile_x
 * consecutive slots form a patch (= remap_apply_args.patch_rows).
 *
 *   row_order_out (device, A.n_rows; required with grid_dims)
 *   patch_ptr     (device) n_patches + 1      n_patches = ceil(n_rows / rows)
 *   patch_ucol    (device) A.nnz (distinct (patch, col) pairs <= nnz)
 *   patch_rowptr  (device) A.n_rows + 1
 *   patch_lidx, patch_val  (device) A.nnz
 *   stats_out     (device) int64[3]: distinct pairs, longest per-patch list
 *                 (patch_umax), most entries in one patch (patch_emax) --
 *                 what the caller needs to size the LDS image (see
 *                 patch_row_bytes) and to decide whether the tile fits
 * Asynchronous on `stream`; nothing is allocated.
 */
REMAP_API
int remap_patches_workspace(int64_t n_rows, int64_t nnz, size_t *bytes_out);
REMAP_API int remap_patches_build(const remap_csr *A, const int64_t *grid_dims,
                        int64_t row_offset, int32_t tile_y, int32_t tile_x,
                        int32_t *row_order_out, int32_t *patch_ptr,
                        int32_t *patch_ucol, int32_t *patch_rowptr,
                        int32_t *patch_lidx, double *patch_val,
                        int64_t *stats_out, void *workspace,
                        size_t workspace_bytes, void *stream);

/*
 * WHICH schedule a mapping gets -- decided (from the measurements recorded in
 * DESIGN.md section 6) and built in one call.  The schedule's arrays live in
 * ONE device arena the caller provides and keeps alive as long as the
 * schedule is used; the struct returns the ready-to-copy fields of
 * remap_apply_args.  Synchronous on `stream` (statistics are read back to
 * choose).  Query both sizes first with remap_schedule_sizes().
 *
 *   family  0: nothing to attach (no destination grid, empty matrix, or no
 *              source-row sharing to exploit)
 *           5: LDS patches   -> row_order, patch_*   (bilinear, coarse->fine)
 *          10: row groups    -> row_order, group_*   (conservative maps;
 *              entry-rich ones -- 2nd-order stencils -- also share_*)
 *           6: plain kernels in a tiled processing order -> row_order only
 *   tune[mode]: what to pass as remap_apply_args.tune for REMAP_MODE_<mode>,
 *              together with REMAP_FLAG_TUNE_HINT (a call the family cannot
 *              serve then falls back by itself)
 *   grid_dims  (HOST) the C-order dims of the WHOLE destination grid, n_dims
 *              = 1 or 2 (0: no grid, nothing is scheduled); row_offset = the
 *              index of A's first row in it (0 unless A is a row shard)
 */
typedef struct remap_schedule {
    int32_t family;
    int32_t entry_rich;          /* >= 10 entries per non-empty row          */
    const int32_t *row_order;    /* (device, arena) or NULL                  */
    const int32_t *patch_ptr;
    const int32_t *patch_ucol;
    const int32_t *patch_rowptr;
    const int32_t *patch_lidx;
    const double *patch_val;
    int32_t patch_rows;
    int32_t patch_umax;
    int32_t patch_emax;
    int32_t patch_row_bytes;
    int64_t n_patches;
    const int64_t *group_meta;
    const int32_t *group_col;
    const double *group_w;
    const int32_t *group_mask;
    const int32_t *group_rid;
    const double *group_frac;
    int64_t n_groups;
    int32_t group_rows;
    int32_t super_tile;          /* 0: groups row-major over the whole grid  */
    int32_t tile_y;              /* patch tile / processing-order tile       */
    int32_t tile_x;
    double ratio;                /* distinct/entries (5), union/entries (10) */
    int64_t n_distinct;          /* distinct pairs (5), union entries (10)   */
    int32_t tune[3][8];
    size_t arena_used;           /* bytes of the arena the schedule occupies */
    /* (ABI 25) entry-rich row groups: the shared union lists of the shared
     * form, remap_apply_args.share_* (NULL / 0 otherwise)                  */
    const int64_t *share_meta;
    const int32_t *share_col;
    const int32_t *share_mask;
    int32_t share_waves;
    int32_t share_reserved;
    int64_t n_share_union;       /* union entries of the supergroups        */
} remap_schedule;

REMAP_API
int remap_schedule_sizes(int64_t n_rows, int64_t nnz, size_t *arena_bytes,
                         size_t *workspace_bytes);
REMAP_API int remap_schedule_auto(const remap_csr *A, const double *frac_b,
                        const int64_t *grid_dims, int32_t n_dims,
                        int64_t row_offset, void *arena, size_t arena_bytes,
                        void *workspace, size_t workspace_bytes,
                        remap_schedule *schedule_out, void *stream);

/*
 * The whole path behind ONE opaque handle, device memory owned by the
 * library (hipMalloc / hipFree) -- for a binder without an allocator of its
 * own; the entry points above never allocate and suit a host layer that has
 * one (pyremap_amd keeps everything in torch tensors).
 *
 * remap_plan_create = `_load_mapping` (remap_numpy.py:72-139): the mapping
 *   file's triplets become the device CSR scipy would build (:134-137:
 *   sorted, duplicates summed), frac_b is copied, and the mapping gets its
 *   kernel schedule (remap_schedule_auto).  Cache the handle where the
 *   reference caches `remapper._matrix`.
 *     row, col (int32, `index_base`-based), S, frac_b: HOST arrays when
 *       host_input != 0 (the library uploads them), device arrays otherwise;
 *       not retained either way
 *     dst_grid_dims (HOST): C-order dims of the destination grid, n_dims =
 *       1 or 2 (0: no grid known -- the plain kernels)
 *   Synchronous on `stream`.  REMAP_ERR_ARG if a triplet is out of range.
 * remap_plan_apply = `_remap_numpy_array` (:223-297): one fused launch over
 *   all rows, asynchronous on `stream`; field layout and modes as in
 *   remap_apply_args (same names, same meaning).  (Two launches writing
 *   disjoint rows for a mapping whose few LONG rows -- more than 96 entries:
 *   the pole caps of a global bilinear map as ESMF makes it -- hold 2 % of
 *   the entries or more: remap_plan_create keeps those rows apart, see
 *   patch_ell_base.)
 *   The calling thread's current HIP device must be the one the plan was
 *   created on (REMAP_ERR_ARG otherwise: the launch would run against
 *   another device's pointers).
 * remap_plan_destroy frees the device memory on the plan's device, whatever
 *   device is current (the caller makes sure no launch still uses it).
 */
typedef struct remap_plan remap_plan;

typedef struct remap_plan_info {
    int64_t n_a;
    int64_t n_b;
    int64_t nnz;                 /* after duplicate summing                  */
    int64_t max_row_nnz;
    int32_t family;              /* remap_schedule.family chosen             */
    int32_t group_rows;
    double ratio;
    size_t device_bytes;         /* device memory the plan holds             */
    int32_t cell_patch_rows;     /* rows per patch of the lanes-across-rows
                                  * plan remap_plan_prepare_short_runs built
                                  * (0: not built)                           */
    int32_t reserved;
} remap_plan_info;

typedef struct remap_field {
    const void *X;               /* (device) source field                    */
    int32_t x_dtype;             /* REMAP_DTYPE_*                            */
    int32_t mode;                /* REMAP_MODE_*                             */
    int64_t n_batch;
    int64_t k_inner;
    int64_t x_row_stride;        /* elements                                 */
    int64_t x_batch_stride;
    double *Y;                   /* (device) float64 result                  */
    int64_t y_row_stride;
    int64_t y_batch_stride;
    double threshold;            /* REMAP_MODE_MASKED                        */
    uint8_t *mask_out;           /* (device) or NULL                         */
    const int32_t *gate;         /* (device) or NULL: see remap_apply_args   */
    int32_t gate_value;
    uint32_t flags;              /* REMAP_FLAG_FMA, REMAP_FLAG_TREE,
                                  * REMAP_FLAG_CELL_MASKS                    */
} remap_field;

REMAP_API int remap_plan_create(int64_t n_b, int64_t n_a, int64_t n_s,
                      const int32_t *row, const int32_t *col, const double *S,
                      int32_t index_base, const double *frac_b,
                      int32_t host_input, const int64_t *dst_grid_dims,
                      int32_t n_dims, void *stream, remap_plan **plan_out);
/*
 * Optional, once per plan, before fields whose contiguous run behind the
 * source axes is short and that come in several batches:
 *   (Time, nCells), MPAS's 2-D time series, the reference's most common
 *   input (tests/test_interpolate.py:57-59; k_inner < 4 and n_batch > 1):
 *   builds the patch plan of the LDS-staged lanes-across-rows kernel (32 x 32
 *   tiles of the destination grid -- 16 x 16 on grids under 128 K cells and
 *   on coarse-to-fine maps of fewer than 1 024 such tiles -- halved until
 *   no patch references more than 2 046 distinct source cells), which remap_plan_apply then uses for such
 *   fields.  Without it they take the unstaged lanes-across-rows kernel
 *   (2.5 x slower on EC30to60 -> 0.5 degree at Time = 120);
 *   (Time, nCells, 4 ... 15 levels) on mappings scheduled as row groups:
 *   two more patch plans -- 16 x 16 tiles for the batch-at-a-time kernel
 *   (4 ... 6 levels), 4 x 8 tiles for the LDS patch kernel (7 ... 15) --
 *   1.3-1.9 x faster than the row groups on such runs.
 * Each plan takes 16 bytes per entry of the mapping.  Allocates device
 * memory (remap_plan_apply never does) and synchronises `stream`; a second
 * call does nothing.
 */
REMAP_API int remap_plan_prepare_short_runs(remap_plan *plan, void *stream);
REMAP_API void remap_plan_destroy(remap_plan *plan);
REMAP_API
int remap_plan_query(const remap_plan *plan, remap_plan_info *info_out);
REMAP_API
int remap_plan_apply(const remap_plan *plan, const remap_field *field,
                     void *stream);

/*
 * `_remap_data_array` in one call (remap_numpy.py:201-204 + :223-297): scan
 * the field for NaNs on the device and enqueue the masked, renormalised
 * branch (threshold = field->threshold) and the frac_b branch, each gated on
 * what the scan finds -- nothing synchronises, nothing is allocated, the
 * sequence is hipGraph-capturable.  What is scanned is what the call reads:
 * the n_a cells x n_batch x k_inner values the field's strides address
 * (`x_elems` is kept for callers of ABI 24 and only checked for sign);
 * `field->mode` and `field->gate` are ignored; `kinds` is a device int32[4]
 * (ABI 25: two more words) the call zeroes and fills (see
 * remap_scan_nan_layout).  On a mapping scheduled as 8-row groups the masked
 * branch is enqueued in each of its forms (REMAP_FLAG_CELL_MASKS,
 * REMAP_FLAG_BATCH_MASKS, neither): four gated launches instead of two.
 */
REMAP_API
int remap_plan_apply_auto(const remap_plan *plan, const remap_field *field,
                          int64_t x_elems, int32_t *kinds, void *stream);

/*
 * OR 1 into *flag (device int32, zeroed by the caller) if any of the n
 * elements of x (device, REMAP_DTYPE_*, element-aligned) is a NaN.
 * Asynchronous on `stream`; the device half of remap_numpy.py:201-204.
 */
REMAP_API
int remap_scan_nan(const void *x, int32_t x_dtype, int64_t n, int32_t *flag,
                   void *stream);

/*
 * remap_scan_nan() with the KIND of the missing values (device int32[2],
 * zeroed by the caller): kinds[0] |= 1 if x holds a NaN (what
 * remap_scan_nan() reports); kinds[1] |= 1 if x holds a NaN, |= 2 if some
 * aligned run of 16 bytes x 64 lanes (128 float64 / 256 float32 elements in
 * memory order) holds NaNs AND numbers -- so kinds[1] is 0 (no NaN), 1 (NaNs
 * in whole runs: whole cells of an (n_a, K) field missing) or 3 (NaNs
 * column by column).  Three gated calls then cover remap_numpy.py:201-204
 * on an entry-rich mapping: FRACB gated on kinds[0] == 0, MASKED with
 * REMAP_FLAG_CELL_MASKS gated on kinds[1] == 1, MASKED without it gated on
 * kinds[1] == 3.  A hint only: results never depend on it.
 */
REMAP_API int remap_scan_nan_kinds(const void *x, int32_t x_dtype, int64_t n,
                         int32_t *kinds, void *stream);

/*
 * The scan with the field's LAYOUT (ABI 25; device int32[4], zeroed by the
 * caller): source cell a's k_inner values of batch b lie at x[b *
 * x_batch_stride + a * x_row_stride + k], as in remap_apply_args.  What is
 * missing is judged per source cell and per batch -- where the cells and the
 * batches are, not in aligned runs of memory, which a (Time, nCells,
 * nVertLevels) field with land cells reads as "column by column":
 *   kinds[0]  1 if the field holds a NaN
 *   kinds[1]  0 no NaN; 1 every cell is missing in ALL of its n_batch *
 *             k_inner columns or in none (land: what REMAP_FLAG_CELL_MASKS
 *             expects); 3 otherwise
 *   kinds[2]  0 no NaN; 1 every batch has the mask of batch 0 (bathymetry:
 *             what REMAP_FLAG_BATCH_MASKS expects); 3 otherwise
 *   kinds[3]  the launch that suits: 0 none (FRACB), 1 MASKED with
 *             REMAP_FLAG_CELL_MASKS, 2 MASKED with REMAP_FLAG_BATCH_MASKS
 *             (n_batch >= 3 only), 3 MASKED without either
 * Four calls gated on kinds[3] == 0 / 1 / 2 / 3 then cover
 * remap_numpy.py:201-204 on an entry-rich mapping without a host round
 * trip.  Hints only: results never depend on them.  Asynchronous on
 * `stream`; kinds[3] is written by a second small launch behind the scan.
 */
REMAP_API int remap_scan_nan_layout(const void *x, int32_t x_dtype,
                          int64_t n_rows, int64_t n_batch, int64_t k_inner,
                          int64_t x_row_stride, int64_t x_batch_stride,
                          int32_t *kinds, void *stream);

/*
 * The two device steps of a ROW SHARD's exchange (no counterpart in the
 * reference, which is single-process; the math is remap_numpy.py:264-268:
 * destination row i reads only the source rows its entries name, so a shard
 * of rows needs X[unique(col[shard])] and nothing else -- whatever the
 * numbering of the source mesh).
 *
 * remap_pack_columns: one-off per shard.  ucols_out (device, capacity
 *   min(nnz, n_cols)) receives the DISTINCT column indices of `col`
 *   ascending, *n_ucols_out (device int64) their number, and col_out (device,
 *   nnz; may alias col) the same entries renumbered to positions in that
 *   list.  The renumbering is monotone: a row's entries keep their order, so
 *   the packed shard reproduces the unsharded result bit for bit.
 *   *bad_out (device int64) counts entries outside [0, n_cols).
 * remap_gather_rows: per batch of fields, on the device that holds them:
 *   dst[b][i][0:row_bytes] = src[b * batch_stride + rows[i] * row_stride ...]
 *   for b < n_batch, i < n_rows; dst is contiguous (n_batch, n_rows,
 *   row_bytes).  With rows = a shard's ucols this is the packed source
 *   buffer that travels to the shard's GPU; strides in BYTES.
 * Both asynchronous on `stream`; nothing is allocated.
 */
REMAP_API int remap_pack_columns_workspace(int64_t n_cols, size_t *bytes_out);
REMAP_API
int remap_pack_columns(const int32_t *col, int64_t nnz, int64_t n_cols,
                       int32_t *col_out, int32_t *ucols_out,
                       int64_t *n_ucols_out, int64_t *bad_out,
                       void *workspace, size_t workspace_bytes, void *stream);
REMAP_API int remap_gather_rows(const void *src, int64_t n_batch,
                      int64_t src_batch_stride_bytes,
                      int64_t src_row_stride_bytes, const int32_t *rows,
                      int64_t n_rows, int64_t row_bytes, void *dst,
                      void *stream);

/*
 * Device-to-device streaming copy of `bytes` (16 B per lane, grid-stride):
 * the box's achievable HBM ceiling, reported beside the roofline numbers.
 */
REMAP_API
int remap_stream_copy(void *dst, const void *src, size_t bytes, void *stream);

/*
 * The shader clock the chip holds right now (asynchronous on `stream`): one
 * wave spins for `micros` microseconds (1 ... 1000) and writes the shader
 * cycles (s_memtime) and the constant-rate 100 MHz ticks (s_memrealtime) it
 * saw go by to ticks_out[0] / ticks_out[1] (device, 2 x int64): MHz = 100 *
 * ticks_out[0] / ticks_out[1].  Queued right behind a series of launches it
 * tells which clock state their times belong to (an instruction-issue-bound
 * kernel such as the LDS patch kernel runs 5.2 or 6.4 ms per launch on
 * config 4 depending on it).  Measurement only.
 */
REMAP_API
int remap_clock_probe(int64_t *ticks_out, int32_t micros, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* REMAP_HIP_H */
