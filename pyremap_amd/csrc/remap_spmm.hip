// remap_spmm.hip -- weight application on MI355X (gfx950, CDNA4).
//
// Replaces the arithmetic of pyremap/remapper/remap_numpy.py:258-278
// (`matrix.dot`, normalisation by frac_b or by the remapped mask, masking)
// and, through strided addressing, the permute/flatten copies of :254-256 and
// :280-295.  See include/remap_hip.h for the contract.
//
// Design (HBM-bound gather; no MFMA -- the contraction is sparse, ~0.17
// flop/byte):
//
//  * rowwave family: one wave64 owns one destination row x one K-chunk.
//    Lanes run ACROSS K (the batched fields), so every access to a source row
//    is a contiguous 16 B-per-lane, 1 KiB-per-wave load (whole 128 B lines),
//    the row's (col, S) pairs are fetched once per wave with one coalesced
//    load and broadcast with v_readlane, and the sum over a row's entries is
//    sequential per lane: no cross-lane reduction, hence the same summation
//    order as scipy's csr_matvecs and bit-identical results when built with
//    -ffp-contract=off (REMAP_FLAG_FMA opts out).
//  * rowscalar family (default): the same decomposition with the row
//    metadata fetched through the scalar cache (wide s_loads into SGPRs), so
//    the vector-memory pipeline only carries X loads and Y stores.
//  * rowgroup family (chosen per mapping by the host): one wave computes 8
//    neighbouring destination rows over the sorted union of their columns,
//    so a source row shared by several of them is loaded once per wave.
//  * patch family (chosen per mapping by the host): LDS-staged gather of
//    each destination patch's distinct source rows by LDS-DMA.
//  * rowlane family (K <= 32): one lane per (row, k); lanes of a wave cover
//    64 / K consecutive rows, X accesses are contiguous over k.
//  * rowcell / patchcell families: lanes ACROSS destination rows, a few
//    fields per lane -- for fields whose K values of one source cell are not
//    contiguous: (Time, nCells), (time, lev, lat, lon), two source axes with
//    other dims between them.  patchcell stages each distinct source cell of
//    a destination patch once in LDS.
//  * Fused epilogue: division by frac_b / by the remapped mask, threshold
//    test, NaN fill and the optional byte mask are applied in registers; the
//    reference's four (n, K) temporaries and its second SpMM never exist.
//  * XCD-aware block map: each XCD (own 4 MiB L2) gets a contiguous range of
//    the chunk-major work list, so the ~nnz/n_a re-touches of a source row by
//    neighbouring destination rows hit that XCD's L2 instead of going back
//    to Infinity Cache / HBM eight times.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include <hip/hip_ext.h>

#include "remap_common.h"

namespace remap {

char *error_buffer()
{
    static thread_local char buf[kErrorBufferSize] = "";
    return buf;
}

namespace {

#include "spmm_device.h"
#include "spmm_rowwave.h"
#include "spmm_patch.h"
#include "spmm_stamps.h"
#include "spmm_rowscalar.h"
#include "spmm_rowgroup.h"
#include "spmm_grouproll.h"
#include "spmm_groupmask.h"
#include "spmm_groupshare.h"
#include "spmm_grouptime.h"
#include "spmm_timeshare.h"
#include "spmm_cellshare.h"
#include "spmm_narrowshare.h"
#include "spmm_rowlane.h"
#include "spmm_rowcell.h"
#include "spmm_patchcell.h"
#include "spmm_rowsub.h"
#include "spmm_strip.h"
#include "spmm_longrow.h"
#include "spmm_longwave.h"

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
typedef void (*kernel_fn)(const KParams, const uint32_t);

template <typename XT, int VEC, int TILES, int UNROLL>
kernel_fn pick_rowwave(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowwave<XT, VEC, TILES, REMAP_MODE_RAW, true, UNROLL>
                   : spmm_rowwave<XT, VEC, TILES, REMAP_MODE_RAW, false, UNROLL>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowwave<XT, VEC, TILES, REMAP_MODE_FRACB, true, UNROLL>
                   : spmm_rowwave<XT, VEC, TILES, REMAP_MODE_FRACB, false, UNROLL>;
    default:
        return fma ? spmm_rowwave<XT, VEC, TILES, REMAP_MODE_MASKED, true, UNROLL>
                   : spmm_rowwave<XT, VEC, TILES, REMAP_MODE_MASKED, false, UNROLL>;
    }
}

template <typename XT>
kernel_fn pick_rowwave_shape(int vec, int tiles, int mode, bool fma)
{
    if (vec == 1)
        return pick_rowwave<XT, 1, 1, 8>(mode, fma);
    switch (tiles) {
    case 1:
        return pick_rowwave<XT, 2, 1, 8>(mode, fma);
    case 2:
        return pick_rowwave<XT, 2, 2, 4>(mode, fma);
    default:
        return pick_rowwave<XT, 2, 4, 2>(mode, fma);
    }
}

template <typename XT, int UNR>
kernel_fn pick_rowlane_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowlane<XT, REMAP_MODE_RAW, true, UNR>
                   : spmm_rowlane<XT, REMAP_MODE_RAW, false, UNR>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowlane<XT, REMAP_MODE_FRACB, true, UNR>
                   : spmm_rowlane<XT, REMAP_MODE_FRACB, false, UNR>;
    default:
        return fma ? spmm_rowlane<XT, REMAP_MODE_MASKED, true, UNR>
                   : spmm_rowlane<XT, REMAP_MODE_MASKED, false, UNR>;
    }
}

// entries fetched together per lane: 4 (config 3's map, us per launch with
// 1 / 4 / 8: K = 1 10.7 / 11.3 / 10.3 -- the launch floor; K = 12 33.9 /
// 28.7 / 30.0; K = 32 70.5 / 59.8 / 67.1; config 1's bilinear map, K = 4:
// 10.6 / 10.6 / 13.3); tune[1] = 1, 4 or 8 overrides
template <typename XT>
kernel_fn pick_rowlane(int unr, int mode, bool fma)
{
    return unr == 1   ? pick_rowlane_mode<XT, 1>(mode, fma)
           : unr == 4 ? pick_rowlane_mode<XT, 4>(mode, fma)
                      : pick_rowlane_mode<XT, 8>(mode, fma);
}

template <typename XT, int TT, int UNR>
kernel_fn pick_rowcell_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowcell<XT, REMAP_MODE_RAW, true, TT, UNR>
                   : spmm_rowcell<XT, REMAP_MODE_RAW, false, TT, UNR>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowcell<XT, REMAP_MODE_FRACB, true, TT, UNR>
                   : spmm_rowcell<XT, REMAP_MODE_FRACB, false, TT, UNR>;
    default:
        return fma ? spmm_rowcell<XT, REMAP_MODE_MASKED, true, TT, UNR>
                   : spmm_rowcell<XT, REMAP_MODE_MASKED, false, TT, UNR>;
    }
}

// tune[1] = fields per lane (4, 8, 16), tune[2] = entries fetched together
template <typename XT>
kernel_fn pick_rowcell(int tt, int unr, int mode, bool fma)
{
    if (tt == 4)
        return unr == 4 ? pick_rowcell_mode<XT, 4, 4>(mode, fma)
                        : pick_rowcell_mode<XT, 4, 2>(mode, fma);
    if (tt == 16)
        return unr == 1 ? pick_rowcell_mode<XT, 16, 1>(mode, fma)
                        : pick_rowcell_mode<XT, 16, 2>(mode, fma);
    return unr == 1   ? pick_rowcell_mode<XT, 8, 1>(mode, fma)
           : unr == 4 ? pick_rowcell_mode<XT, 8, 4>(mode, fma)
                      : pick_rowcell_mode<XT, 8, 2>(mode, fma);
}

template <typename XT, int SUB, bool TREE>
kernel_fn pick_rowsub_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowsub<XT, REMAP_MODE_RAW, true, SUB, TREE>
                   : spmm_rowsub<XT, REMAP_MODE_RAW, false, SUB, TREE>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowsub<XT, REMAP_MODE_FRACB, true, SUB, TREE>
                   : spmm_rowsub<XT, REMAP_MODE_FRACB, false, SUB, TREE>;
    default:
        return fma ? spmm_rowsub<XT, REMAP_MODE_MASKED, true, SUB, TREE>
                   : spmm_rowsub<XT, REMAP_MODE_MASKED, false, SUB, TREE>;
    }
}

template <typename XT>
kernel_fn pick_rowsub(int sub, bool tree, int mode, bool fma)
{
    if (sub == 4)
        return tree ? pick_rowsub_mode<XT, 4, true>(mode, fma)
                    : pick_rowsub_mode<XT, 4, false>(mode, fma);
    return tree ? pick_rowsub_mode<XT, 8, true>(mode, fma)
                : pick_rowsub_mode<XT, 8, false>(mode, fma);
}

typedef void (*patch_fn)(const KParams, const uint32_t, const int32_t *,
                         const double *, const int32_t *, const int32_t *,
                         const int32_t *, const int32_t *, const double *,
                         const int32_t, const int32_t, const int32_t,
                         const int64_t);

template <typename XT, int WC, bool DMA, int BLOCK = kPatchBlock>
patch_fn pick_patch_wc(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_patch<XT, REMAP_MODE_RAW, true, WC, DMA, BLOCK>
                   : spmm_patch<XT, REMAP_MODE_RAW, false, WC, DMA, BLOCK>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_patch<XT, REMAP_MODE_FRACB, true, WC, DMA, BLOCK>
                   : spmm_patch<XT, REMAP_MODE_FRACB, false, WC, DMA, BLOCK>;
    default:
        return fma ? spmm_patch<XT, REMAP_MODE_MASKED, true, WC, DMA, BLOCK>
                   : spmm_patch<XT, REMAP_MODE_MASKED, false, WC, DMA, BLOCK>;
    }
}

// wc = columns per K-chunk (128 / 64); dma = 16-byte LDS-DMA pieces of
// float64 rows (else through registers, converting f32 on the way); block =
// threads per workgroup (1 024; 512 for 64-column chunks only)
patch_fn pick_patch(bool f32, int mode, bool fma, int wc, bool dma,
                    int block = kPatchBlock)
{
    if (block == 512 && wc == 64) {
        if (f32)
            return pick_patch_wc<float, 64, false, 512>(mode, fma);
        return dma ? pick_patch_wc<double, 64, true, 512>(mode, fma)
                   : pick_patch_wc<double, 64, false, 512>(mode, fma);
    }
    if (f32)
        return wc == 64 ? pick_patch_wc<float, 64, false>(mode, fma)
                        : pick_patch_wc<float, 128, false>(mode, fma);
    if (!dma)
        return wc == 64 ? pick_patch_wc<double, 64, false>(mode, fma)
                        : pick_patch_wc<double, 128, false>(mode, fma);
    return wc == 64 ? pick_patch_wc<double, 64, true>(mode, fma)
                    : pick_patch_wc<double, 128, true>(mode, fma);
}

typedef void (*cell_fn)(const KParams, const uint32_t, const int32_t *,
                        const double *, const int32_t *, const int32_t *,
                        const int32_t *, const int32_t *, const double *,
                        const int32_t, const int32_t, const int64_t,
                        const int64_t *);

template <typename XT, int TT, int LAYOUT>
cell_fn pick_patchcell_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_patchcell<XT, REMAP_MODE_RAW, true, TT, LAYOUT>
                   : spmm_patchcell<XT, REMAP_MODE_RAW, false, TT, LAYOUT>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_patchcell<XT, REMAP_MODE_FRACB, true, TT, LAYOUT>
                   : spmm_patchcell<XT, REMAP_MODE_FRACB, false, TT, LAYOUT>;
    default:
        return fma ? spmm_patchcell<XT, REMAP_MODE_MASKED, true, TT, LAYOUT>
                   : spmm_patchcell<XT, REMAP_MODE_MASKED, false, TT, LAYOUT>;
    }
}

template <typename XT, int LAYOUT>
cell_fn pick_patchcell_tt(int tt, int mode, bool fma)
{
    return tt == 4    ? pick_patchcell_mode<XT, 4, LAYOUT>(mode, fma)
           : tt == 16 ? pick_patchcell_mode<XT, 16, LAYOUT>(mode, fma)
                      : pick_patchcell_mode<XT, 8, LAYOUT>(mode, fma);
}

template <typename XT, int TT, int BLOCK>
cell_fn pick_patchtime_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_patchtime<XT, REMAP_MODE_RAW, true, TT, BLOCK>
                   : spmm_patchtime<XT, REMAP_MODE_RAW, false, TT, BLOCK>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_patchtime<XT, REMAP_MODE_FRACB, true, TT, BLOCK>
                   : spmm_patchtime<XT, REMAP_MODE_FRACB, false, TT, BLOCK>;
    default:
        return fma ? spmm_patchtime<XT, REMAP_MODE_MASKED, true, TT, BLOCK>
                   : spmm_patchtime<XT, REMAP_MODE_MASKED, false, TT, BLOCK>;
    }
}

template <typename XT, int BLOCK>
cell_fn pick_patchtime_tt(int tt, int mode, bool fma)
{
    return tt == 2   ? pick_patchtime_mode<XT, 2, BLOCK>(mode, fma)
           : tt == 4 ? pick_patchtime_mode<XT, 4, BLOCK>(mode, fma)
                     : pick_patchtime_mode<XT, 8, BLOCK>(mode, fma);
}

// (Time, nCells, 4 ... 15): a batch at a time, 4 or 8 columns per chunk
template <typename XT, int BLOCK, int TT>
cell_fn pick_patchruns_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_patchtime<XT, REMAP_MODE_RAW, true, TT, BLOCK, true>
                   : spmm_patchtime<XT, REMAP_MODE_RAW, false, TT, BLOCK,
                                    true>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_patchtime<XT, REMAP_MODE_FRACB, true, TT, BLOCK,
                                    true>
                   : spmm_patchtime<XT, REMAP_MODE_FRACB, false, TT, BLOCK,
                                    true>;
    default:
        return fma ? spmm_patchtime<XT, REMAP_MODE_MASKED, true, TT, BLOCK,
                                    true>
                   : spmm_patchtime<XT, REMAP_MODE_MASKED, false, TT, BLOCK,
                                    true>;
    }
}

// (8 columns per chunk -- one chunk per batch at L = 8 -- was built and
// measured in round 5: slower at every L, 0.63 against 0.59 ms at L = 8, 0.83
// against 0.68 at L = 10: twice the LDS image, half the workgroups per CU)
template <typename XT, int TT>
cell_fn pick_patchruns_block(int mode, bool fma, int block)
{
    return block == 256 ? pick_patchruns_mode<XT, 256, TT>(mode, fma)
                        : pick_patchruns_mode<XT, 512, TT>(mode, fma);
}

template <typename XT>
cell_fn pick_patchruns(int mode, bool fma, int block, int tt)
{
    // (8 / 10 / 12 columns per chunk -- the whole batch of 8 ... 12 levels
    // one chunk, one LDS image -- were built and measured in round 5: no
    // better than 4 per chunk and behind the 4 x 8 LDS patches at every such
    // L; removed)
    return tt == 6 ? pick_patchruns_block<XT, 6>(mode, fma, block)
                   : pick_patchruns_block<XT, 4>(mode, fma, block);
}

template <typename XT>
cell_fn pick_patchtime(int tt, int mode, bool fma, int block)
{
    return block == 256   ? pick_patchtime_tt<XT, 256>(tt, mode, fma)
           : block == 512 ? pick_patchtime_tt<XT, 512>(tt, mode, fma)
                          : pick_patchtime_tt<XT, 1024>(tt, mode, fma);
}

template <typename XT>
cell_fn pick_patchcell(int tt, int mode, bool fma, int layout)
{
    // long rows (column-major entries): fewer fields per lane -- a row is
    // ONE dependent chain of hundreds of entries, and a lone wave per SIMD
    // pays every instruction of it in full: short steps, more workgroups
    if (layout == 1 && tt == 1)
        return pick_patchcell_mode<XT, 1, 1>(mode, fma);
    if (layout == 1 && tt == 2)
        return pick_patchcell_mode<XT, 2, 1>(mode, fma);
    return layout == 1 ? pick_patchcell_tt<XT, 1>(tt, mode, fma)
                       : pick_patchcell_tt<XT, 0>(tt, mode, fma);
}

// LDS a workgroup may ask for and still leave room for a second one per CU
constexpr uint32_t kPatchLdsMax = 160 * 1024;

bool patch_usable(const remap_apply_args *a, int64_t K64)
{
    return a->patch_ptr && a->patch_ucol && a->patch_lidx &&
           a->patch_rowptr && a->patch_val && a->patch_rows > 0 &&
           a->patch_rows < kPatchBlock && a->n_patches > 0 &&
           a->patch_umax >= 0 && a->patch_emax >= 0 && K64 >= 2 &&
           (a->patch_row_bytes == 1024 || a->patch_row_bytes == 512) &&
           patch_lds_bytes(a->patch_umax, a->patch_emax, a->patch_rows,
                           a->patch_row_bytes) <= kPatchLdsMax &&
           a->row_end - a->row_begin <=
               a->n_patches * (int64_t)a->patch_rows &&
           a->row_end - a->row_begin >
               (a->n_patches - 1) * (int64_t)a->patch_rows;
}

template <typename XT>
struct ScalarFn {
    typedef void (*type)(const KParams, const uint32_t, const int64_t *,
                         const int32_t *, const double *, const int32_t *,
                         const double *, const XT *);
};

template <typename XT, int VEC, int TILES>
typename ScalarFn<XT>::type pick_rowscalar_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowscalar<XT, VEC, TILES, REMAP_MODE_RAW, true>
                   : spmm_rowscalar<XT, VEC, TILES, REMAP_MODE_RAW, false>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowscalar<XT, VEC, TILES, REMAP_MODE_FRACB, true>
                   : spmm_rowscalar<XT, VEC, TILES, REMAP_MODE_FRACB, false>;
    default:
        return fma ? spmm_rowscalar<XT, VEC, TILES, REMAP_MODE_MASKED, true>
                   : spmm_rowscalar<XT, VEC, TILES, REMAP_MODE_MASKED, false>;
    }
}

template <typename XT>
typename ScalarFn<XT>::type pick_rowscalar(int vec, int tiles, int mode,
                                           bool fma)
{
    if (vec == 1)
        return tiles == 1 ? pick_rowscalar_mode<XT, 1, 1>(mode, fma)
                          : pick_rowscalar_mode<XT, 1, 2>(mode, fma);
    return tiles == 1 ? pick_rowscalar_mode<XT, 2, 1>(mode, fma)
                      : pick_rowscalar_mode<XT, 2, 2>(mode, fma);
}

template <typename XT>
int launch_rowscalar(const remap_apply_args *a, const KParams &p, int vec,
                     int tiles, bool fma, int64_t grid, hipStream_t stream)
{
    typename ScalarFn<XT>::type fn =
        pick_rowscalar<XT>(vec, tiles, a->mode, fma);
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)), dim3(kBlock), 0,
                       stream, p, a->flags, a->A.rowptr, a->A.col, a->A.val,
                       a->row_order, a->frac_b,
                       static_cast<const XT *>(a->X));
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// Diagnostic build only (tune[7]): KiB of unused dynamic LDS per block, an
// occupancy throttle for bottleneck experiments.  The product build launches
// with no dynamic LDS.
hipError_t diag_lds_throttle(const remap_apply_args *a, const void *fn,
                             uint32_t &lds_bytes)
{
#ifdef REMAP_DIAG
    if (a->tune[7] > 0 && (uint32_t)a->tune[7] * 1024u > lds_bytes)
        lds_bytes = a->tune[7] * 1024u;
    if (lds_bytes > 64 * 1024)
        return hipFuncSetAttribute(
            fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
#else
    (void)a;
    (void)fn;
    (void)lds_bytes;
#endif
    return hipSuccess;
}

template <typename XT>
struct GroupFn {
    typedef void (*type)(const KParams, const uint32_t, const int64_t *,
                         const int32_t *, const double *, const int32_t *,
                         const int32_t *, const double *, const XT *);
};

template <typename XT, int TILES, int G, int UNR, int VEC>
typename GroupFn<XT>::type pick_rowgroup_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowgroup<XT, TILES, REMAP_MODE_RAW, true, G, UNR,
                                   VEC>
                   : spmm_rowgroup<XT, TILES, REMAP_MODE_RAW, false, G, UNR,
                                   VEC>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowgroup<XT, TILES, REMAP_MODE_FRACB, true, G, UNR,
                                   VEC>
                   : spmm_rowgroup<XT, TILES, REMAP_MODE_FRACB, false, G, UNR,
                                   VEC>;
    default:
        return fma ? spmm_rowgroup<XT, TILES, REMAP_MODE_MASKED, true, G, UNR,
                                   VEC>
                   : spmm_rowgroup<XT, TILES, REMAP_MODE_MASKED, false, G,
                                   UNR, VEC>;
    }
}

template <typename XT, int G, int UNR>
typename GroupFn<XT>::type pick_rowgroup_tiles(int tiles, int mode, bool fma)
{
    return tiles == 1 ? pick_rowgroup_mode<XT, 1, G, UNR, 2>(mode, fma)
                      : pick_rowgroup_mode<XT, 2, G, UNR, 2>(mode, fma);
}

// vec = elements per lane and tile: 2 (16-byte accesses; even strides and
// level counts, 16-byte aligned bases), or 1 for everything else -- (Time,
// nCells, nVertLevelsP1 = 61), L137, a view that starts at an odd element:
// two tiles of 64 columns, so a wave still covers 128 columns per entry.
// (Keeping two elements per lane there with element-aligned 16-byte
// accesses -- pairs cut per batch, the odd last column fetched one element
// early -- was built and measured: no faster than this, the misaligned wide
// accesses are split by the memory pipeline; removed.)
template <typename XT, int G>
typename GroupFn<XT>::type pick_rowgroup_shape(int unr, int tiles, int vec,
                                               int mode, bool fma)
{
    if (vec == 1)
        return tiles == 1 ? pick_rowgroup_mode<XT, 1, G, 8, 1>(mode, fma)
                          : pick_rowgroup_mode<XT, 2, G, 8, 1>(mode, fma);
    return unr == 4    ? pick_rowgroup_tiles<XT, G, 4>(tiles, mode, fma)
           : unr == 16 ? pick_rowgroup_tiles<XT, G, 16>(tiles, mode, fma)
                       : pick_rowgroup_tiles<XT, G, 8>(tiles, mode, fma);
}

#ifdef REMAP_DIAG
// the lock-step experiment (diagnostic build only; tools/lockstep.py,
// profiles/r03_analysis/lockstep.md): float64, 2 x 2 groups, one tile
template <int BLOCK>
GroupFn<double>::type pick_rowgroup_lock(int mode)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return spmm_rowgroup<double, 1, REMAP_MODE_RAW, false, 4, 8, 2, true,
                             BLOCK>;
    case REMAP_MODE_FRACB:
        return spmm_rowgroup<double, 1, REMAP_MODE_FRACB, false, 4, 8, 2,
                             true, BLOCK>;
    default:
        return spmm_rowgroup<double, 1, REMAP_MODE_MASKED, false, 4, 8, 2,
                             true, BLOCK>;
    }
}
#endif

// the rolling form (spmm_grouproll.h): float64, two elements per lane
template <int TILES, int G, int UNR>
GroupFn<double>::type pick_grouproll_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_grouproll<double, TILES, REMAP_MODE_RAW, true, G,
                                    UNR, 2>
                   : spmm_grouproll<double, TILES, REMAP_MODE_RAW, false, G,
                                    UNR, 2>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_grouproll<double, TILES, REMAP_MODE_FRACB, true, G,
                                    UNR, 2>
                   : spmm_grouproll<double, TILES, REMAP_MODE_FRACB, false, G,
                                    UNR, 2>;
    default:
        return fma ? spmm_grouproll<double, TILES, REMAP_MODE_MASKED, true, G,
                                    UNR, 2>
                   : spmm_grouproll<double, TILES, REMAP_MODE_MASKED, false,
                                    G, UNR, 2>;
    }
}

template <int G>
GroupFn<double>::type pick_grouproll(int unr, int tiles, int mode, bool fma)
{
    if (unr == 6)
        return tiles == 1 ? pick_grouproll_mode<1, G, 6>(mode, fma)
                          : pick_grouproll_mode<2, G, 6>(mode, fma);
    return tiles == 1 ? pick_grouproll_mode<1, G, 8>(mode, fma)
                      : pick_grouproll_mode<2, G, 8>(mode, fma);
}

template <typename XT>
int launch_rowgroup(const remap_apply_args *a, const KParams &p, int tiles,
                    int unr, int vec, int wpb, bool fma, int64_t grid,
                    hipStream_t stream, bool lock = false, int roll = 0,
                    bool cell_masks = false)
{
    typename GroupFn<XT>::type fn =
        a->group_rows == 8
            ? pick_rowgroup_shape<XT, 8>(unr, tiles, vec, a->mode, fma)
            : pick_rowgroup_shape<XT, 4>(unr, tiles, vec, a->mode, fma);
    if constexpr (std::is_same<XT, double>::value) {
        if (a->group_rows == 16)   // 2 x 8 tiles: float64, 2 per lane
            fn = unr == 4 ? pick_rowgroup_tiles<double, 16, 4>(tiles, a->mode,
                                                               fma)
                          : pick_rowgroup_tiles<double, 16, 8>(tiles, a->mode,
                                                               fma);
        if (roll && vec == 2)
            fn = a->group_rows == 16
                     ? pick_grouproll<16>(roll, tiles, a->mode, fma)
                 : a->group_rows == 8
                     ? pick_grouproll<8>(roll, tiles, a->mode, fma)
                     : pick_grouproll<4>(roll, tiles, a->mode, fma);
    }
    // REMAP_FLAG_CELL_MASKS, masked mode on 8-row groups: per-row
    // normalisers while the validity of a source cell is the same in all of
    // a wave's columns (spmm_groupmask.h), two K tiles per wave; float32
    // fields too (their per-lane form holds 170 VGPRs: 2 waves per SIMD)
    if (cell_masks && !roll)
        fn = fma ? spmm_groupmask<XT, 2, true, 8, 8, 2>
                 : spmm_groupmask<XT, 2, false, 8, 8, 2>;
#ifdef REMAP_DIAG
    if constexpr (std::is_same<XT, double>::value) {
        if (lock)
            fn = wpb > 4 ? pick_rowgroup_lock<1024>(a->mode)
                         : pick_rowgroup_lock<kBlock>(a->mode);
    }
#else
    (void)lock;
#endif
    uint32_t lds_bytes = 0;
    REMAP_HIP_CHECK(diag_lds_throttle(a, reinterpret_cast<const void *>(fn),
                                      lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)),
                       dim3(kWave * wpb), lds_bytes, stream, p, a->flags,
                       a->group_meta,
                       a->group_col, a->group_w, a->group_mask, a->group_rid,
                       a->group_frac, static_cast<const XT *>(a->X));
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// REMAP_FLAG_BATCH_MASKS, masked mode on 8-row groups: time-major columns,
// one normaliser per lane and row (spmm_grouptime.h)
template <typename XT>
int launch_grouptime(const remap_apply_args *a, const KParams &p, int wpb,
                     bool fma, int64_t grid, hipStream_t stream)
{
    typename GroupFn<XT>::type fn = fma ? spmm_grouptime<XT, true, 8, 8, 4>
                                        : spmm_grouptime<XT, false, 8, 8, 4>;
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)),
                       dim3(kWave * wpb), 0, stream, p, a->flags,
                       a->group_meta, a->group_col, a->group_w, a->group_mask,
                       a->group_rid, a->group_frac,
                       static_cast<const XT *>(a->X));
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// ... and through the LDS ring of the shared form where the mapping has the
// shared lists (spmm_timeshare.h): float64 fields in whole 16-byte pieces
int launch_timeshare(const remap_apply_args *a, const KParams &p, bool fma,
                     int64_t grid, hipStream_t stream)
{
    void (*fn)(const KParams, const uint32_t, const int64_t *,
               const int32_t *, const double *, const int32_t *,
               const int32_t *, const int64_t *, const int32_t *,
               const int32_t *, const double *) =
        fma ? spmm_timeshare<true, 1> : spmm_timeshare<false, 1>;
    // the ring: two buffers of 8 entries x (4 slices x 512 B); 2 x 4 slots
    // of a step's weights; slack
    uint32_t lds_bytes = 2u * (8u * 2048u + 4u * 512u) + 512u;
    REMAP_HIP_CHECK(diag_lds_throttle(a, reinterpret_cast<const void *>(fn),
                                      lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)), dim3(kWave * 4),
                       lds_bytes, stream, p, a->flags, a->group_meta,
                       a->group_col, a->group_w, a->group_mask, a->group_rid,
                       a->share_meta, a->share_col, a->share_mask,
                       static_cast<const double *>(a->X));
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// REMAP_FLAG_CELL_MASKS through the LDS ring of the shared form: one
// normaliser per ROW (spmm_cellshare.h); float64 fields in whole 16-byte
// pieces, 256 columns per workgroup
int launch_cellshare(const remap_apply_args *a, const KParams &p, bool fma,
                     int64_t grid, hipStream_t stream)
{
    void (*fn)(const KParams, const uint32_t, const int64_t *,
               const int32_t *, const double *, const int32_t *,
               const int32_t *, const int64_t *, const int32_t *,
               const int32_t *, const double *) =
        fma ? spmm_cellshare<true, 1> : spmm_cellshare<false, 1>;
    // the ring: two buffers of 8 entries x 2 KiB; 2 x 4 slots of a step's
    // weights; slack
    uint32_t lds_bytes = 2u * (8u * 2048u + 4u * 512u) + 512u;
    REMAP_HIP_CHECK(diag_lds_throttle(a, reinterpret_cast<const void *>(fn),
                                      lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)), dim3(kWave * 4),
                       lds_bytes, stream, p, a->flags, a->group_meta,
                       a->group_col, a->group_w, a->group_mask, a->group_rid,
                       a->share_meta, a->share_col, a->share_mask,
                       static_cast<const double *>(a->X));
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// the shared form (spmm_groupshare.h): float64, two elements per lane
typedef void (*share_fn)(const KParams, const uint32_t, const int64_t *,
                         const double *, const int32_t *, const double *,
                         const int64_t *, const int32_t *, const int32_t *,
                         const double *);

// (built, measured on config 5 and NOT instantiated here -- the template
// keeps the parameters: 2-wave workgroups over 4 x 4 tiles, 28.5 ms against
// 20.5; rings of 8 x 3, 4 x 3, 4 x 4 entries x buffers, 31.2 / 21.4 / 21.2;
// the masked mode with per-lane normalisers, 32.1 at one K tile and 37.8 at
// two against 27.0 of the 8-row groups: profiles/r06_analysis/
// config5_share.md.  The masked mode's shared form is spmm_timeshare.)
template <int TILES, bool FMA>
share_fn pick_groupshare(int mode)
{
    return mode == REMAP_MODE_RAW
               ? spmm_groupshare<TILES, REMAP_MODE_RAW, FMA, 4, 8, 2, 2>
               : spmm_groupshare<TILES, REMAP_MODE_FRACB, FMA, 4, 8, 2, 2>;
}

int launch_groupshare(const remap_apply_args *a, const KParams &p, int tiles,
                      bool fma, int64_t grid, hipStream_t stream)
{
    share_fn fn = tiles == 1 ? (fma ? pick_groupshare<1, true>(a->mode)
                                    : pick_groupshare<1, false>(a->mode))
                             : (fma ? pick_groupshare<2, true>(a->mode)
                                    : pick_groupshare<2, false>(a->mode));
    // the ring: two buffers of 8 entries, 1 KiB per entry and K tile; 2 x 4
    // slots of a step's weights; slack for the lanes that read past the last
    // slot
    uint32_t lds_bytes =
        2u * (8u * 1024u * static_cast<uint32_t>(tiles) + 4u * 512u) + 512u;
    REMAP_HIP_CHECK(diag_lds_throttle(a, reinterpret_cast<const void *>(fn),
                                      lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)),
                       dim3(kWave * 4), lds_bytes, stream, p, a->flags,
                       a->group_meta, a->group_w, a->group_rid, a->group_frac,
                       a->share_meta, a->share_col, a->share_mask,
                       static_cast<const double *>(a->X));
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// the shared form for at most 64 columns (spmm_narrowshare.h)
int launch_narrowshare(const remap_apply_args *a, const KParams &p, bool fma,
                       int64_t grid, hipStream_t stream)
{
    share_fn fn =
        a->mode == REMAP_MODE_RAW
            ? (fma ? spmm_narrowshare<REMAP_MODE_RAW, true, 2>
                   : spmm_narrowshare<REMAP_MODE_RAW, false, 2>)
            : (fma ? spmm_narrowshare<REMAP_MODE_FRACB, true, 2>
                   : spmm_narrowshare<REMAP_MODE_FRACB, false, 2>);
    // the ring: two buffers of 8 entries x 512 B; 2 x 4 slots of a step's
    // weights; slack
    uint32_t lds_bytes = 2u * (8u * 512u + 4u * 512u) + 512u;
    REMAP_HIP_CHECK(diag_lds_throttle(a, reinterpret_cast<const void *>(fn),
                                      lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)),
                       dim3(kWave * 4), lds_bytes, stream, p, a->flags,
                       a->group_meta, a->group_w, a->group_rid, a->group_frac,
                       a->share_meta, a->share_col, a->share_mask,
                       static_cast<const double *>(a->X));
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

bool aligned(const void *p, size_t a)
{
    return (reinterpret_cast<uintptr_t>(p) % a) == 0;
}

// ---------------------------------------------------------------------------
// dispatch: validate -> describe the call -> pick a family -> shape the grid
// ---------------------------------------------------------------------------

// What the launch helpers need to know about one call.
struct Call {
    int64_t K;            // flat columns = n_batch * k_inner
    int64_t n_rows;       // row_end - row_begin
    bool fma;
    bool f32;
    bool can_vec2;        // two elements per lane: even strides, aligned bases
    bool dma16;           // 16-byte pieces of X rows are aligned and whole
    bool small_offsets;   // byte offsets inside a row fit 32 bits
    bool patch_ok;        // a usable patch plan is attached
    bool cell_ok;         // a patch plan family 7 can use is attached
    bool group_ok;        // a usable row-group schedule is attached
    bool share_ok;        // ... and shared union lists on top of it
    bool strip_ok;        // a strip schedule this call can run on is attached
};

// Argument checks.  Returns REMAP_OK with c.K == 0 for an empty output.
int check_args(const remap_apply_args *a, Call &c)
{
    if (!a)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: args is NULL");
    const remap_csr &A = a->A;
    if (A.n_rows < 0 || A.n_cols < 0 || A.nnz < 0)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: negative CSR size");
    if (a->row_begin < 0 || a->row_end > A.n_rows ||
        a->row_begin > a->row_end)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: rows [%lld, %lld) outside [0, %lld)",
                    (long long)a->row_begin, (long long)a->row_end,
                    (long long)A.n_rows);
    if (a->n_batch < 0 || a->k_inner < 0)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: negative batch size");
    c.K = a->n_batch * a->k_inner;
    c.n_rows = a->row_end - a->row_begin;
    if (c.n_rows == 0 || c.K == 0) {
        c.K = 0;
        return REMAP_OK;  // empty output: nothing to launch
    }
    if (!A.rowptr || !a->Y || (!a->X && A.n_cols > 0))
        return fail(REMAP_ERR_ARG, "remap_apply_f64: NULL device pointer");
    if (A.nnz > 0 && (!A.col || !A.val))
        return fail(REMAP_ERR_ARG, "remap_apply_f64: NULL col/val");
    if (a->x_dtype != REMAP_DTYPE_F64 && a->x_dtype != REMAP_DTYPE_F32)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: unknown x_dtype %d",
                    a->x_dtype);
    if (a->mode < REMAP_MODE_RAW || a->mode > REMAP_MODE_MASKED)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: unknown mode %d",
                    a->mode);
    if (a->mode == REMAP_MODE_FRACB && !a->frac_b)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: REMAP_MODE_FRACB needs frac_b");
    if (a->x_src_fold < 0 || a->x_src_fold >= (int64_t(1) << 31) ||
        (a->x_src_fold != 0 &&
         (a->x_outer_stride < 0 || A.n_cols % a->x_src_fold != 0)))
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: x_src_fold = %lld does not divide the "
                    "%lld source cells", (long long)a->x_src_fold,
                    (long long)A.n_cols);
    if (a->x_src_fold != 0 && a->tune[0] != 0 && a->tune[0] != 4 &&
        a->tune[0] != 7 && a->tune[0] != 9 && a->tune[0] != 11 &&
        !(a->flags & REMAP_FLAG_TUNE_HINT))
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: two non-adjacent source axes "
                    "(x_src_fold) are served by kernel families 4, 7, 9 "
                    "and 11");
    if (c.K >= (int64_t(1) << 31))
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: K = %lld fields per call exceeds 2^31",
                    (long long)c.K);
#ifndef REMAP_DIAG
    if (a->tune[6] != 0 || a->tune[7] != 0)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: tune[6] / tune[7] are switches of the "
                    "diagnostic build (-DREMAP_DIAG), absent from this one");
#endif
    c.fma = (a->flags & REMAP_FLAG_FMA) != 0;
    c.f32 = a->x_dtype == REMAP_DTYPE_F32;
    const size_t xelem = c.f32 ? 4 : 8;
    c.can_vec2 =
        (a->k_inner % 2 == 0) && (a->x_row_stride % 2 == 0) &&
        (a->x_batch_stride % 2 == 0) && (a->y_row_stride % 2 == 0) &&
        (a->y_batch_stride % 2 == 0) && aligned(a->X, 2 * xelem) &&
        aligned(a->Y, 16);
    // LDS-DMA moves 16 bytes (2 doubles) per lane
    c.dma16 = c.can_vec2 && !c.f32;
    // per-lane byte offset of the last flat column, from a row base
    c.small_offsets =
        ((a->n_batch - 1) * a->x_batch_stride + a->k_inner) *
            (int64_t)xelem < (int64_t(1) << 31);
    c.patch_ok = patch_usable(a, c.K);
    c.cell_ok = a->patch_ptr && a->patch_ucol && a->patch_lidx &&
                a->patch_rowptr && a->patch_val && a->patch_rows > 0 &&
                a->n_patches > 0 && a->patch_umax >= 0 &&
                // the LDS image run_patchcell asks for at its smallest TT:
                // its pitch is the padded list length, not umax
                (int64_t)((a->patch_umax + 2) & ~1) *
                        (a->patch_ell_base ? 1 : 4) * 8 <=
                    (int64_t)kPatchLdsMax &&
                c.n_rows <= a->n_patches * (int64_t)a->patch_rows &&
                c.n_rows > (a->n_patches - 1) * (int64_t)a->patch_rows;
    c.group_ok = a->group_meta && a->group_col && a->group_w &&
                 a->group_mask && a->group_rid && a->group_frac &&
                 (a->group_rows == 8 || a->group_rows == 4 ||
                  (a->group_rows == 16 && !c.f32 && c.can_vec2)) &&
                 a->group_reserved == 0 &&
                 a->n_groups ==
                     (c.n_rows + a->group_rows - 1) / a->group_rows;
    c.share_ok = c.group_ok && a->group_rows == 8 && a->share_meta &&
                 a->share_col && a->share_mask &&
                 (a->share_waves == 2 || a->share_waves == 4) &&
                 a->share_reserved == 0;
    const remap_strips *st = a->strips;
    c.strip_ok = st && st->n_units > 0 && st->steps_per_unit > 0 &&
                 st->rows_per_wave > 0 && st->ring_slots >= 2 &&
                 st->ring_slots % 2 == 0 &&
                 (int64_t)st->ring_slots * kStripRowBytes <=
                     (int64_t)kPatchLdsMax &&
                 st->depth >= 1 && st->depth <= 6 &&
                 st->meta_slot_bytes > 0 && st->meta_slot_bytes % 1024 == 0 &&
                 st->waves >= 1 && st->waves + st->depth <= 16 &&
                 (int64_t)(st->ring_slots + 2) * kStripRowBytes +
                         (int64_t)(st->depth + 1) * st->meta_slot_bytes +
                         256 <= (int64_t)kPatchLdsMax &&
                 st->unit_steps && st->arr_ptr && st->arr_src &&
                 st->arr_slot && st->meta_ptr && st->meta &&
                 // the whole mapping, float64, one batch of contiguous
                 // columns in whole 16-byte pieces
                 a->row_begin == 0 && a->row_end == A.n_rows && !c.f32 &&
                 a->n_batch == 1 && c.dma16 && a->x_src_fold == 0;
    return REMAP_OK;
}

KParams base_params(const remap_apply_args *a, const Call &c)
{
    KParams p;
    p.rowptr = a->A.rowptr;
    p.col = a->A.col;
    p.val = a->A.val;
    p.X = a->X;
    p.Y = a->Y;
    p.frac_b = a->frac_b;
    p.mask_out = a->mask_out;
    p.row_order = a->row_order;
    p.gate = a->gate;
    p.gate_value = a->gate_value;
#ifdef REMAP_DIAG
    p.diag = a->tune[6];
#endif
    p.x_range = p.y_range = 0;
    p.row_begin = a->row_begin;
    p.row_end = a->row_end;
    p.ldx = a->x_row_stride;
    p.bsx = a->x_batch_stride;
    p.src_fold = static_cast<uint32_t>(a->x_src_fold);
    p.src_outer = a->x_outer_stride;
    p.ldy = a->y_row_stride;
    p.bsy = a->y_batch_stride;
    p.thr = a->threshold;
    p.K = static_cast<uint32_t>(c.K);
    p.k_inner = static_cast<uint32_t>(a->k_inner);
    p.bpc = 0;
    p.n_rowblocks = p.n_blocks = p.blocks_per_xcd = 0;
    p.rows_per_wave = 0;
    p.xcd_map = 0;
    p.x_pairs = 0;
    p.y_pairs = 0;
    return p;
}

// Fields whose contiguous run behind the source axes is at most 3 elements
// and that come in several batches -- (Time, nCells), (Time, nCells, 3) --
// are served by the lanes-across-rows kernels.  Measured on config 3's map,
// (T, nCells, L) with T * L = 512, fraction of 8 TB/s, patchcell / rowgroup:
// L = 2 0.351 / 0.070, 3 0.230 / 0.096, 4 0.171 / 0.186, 6 0.096 / 0.192,
// 7 0.079 / 0.176, 8 - / 0.387: from 4 elements on the lanes-across-K
// kernels win (the 8 fields a lane of patchcell owns then straddle batches
// and its 8-byte Y stores lie 8 * L bytes apart).
bool short_runs(const remap_apply_args *a)
{
    // (two non-adjacent source axes: only the lanes-across-rows kernels
    // address a cell through two strides)
    return (a->k_inner < 4 && a->n_batch > 1) || a->x_src_fold != 0;
}

// family 7 with tune[2] = 2: (Time, nCells, 4 ... 15) a batch at a time, the
// results written out through LDS (spmm_patchtime<..., RUNS>)
bool runs_usable(const remap_apply_args *a, const Call &c)
{
    if (!c.cell_ok || a->patch_ell_base || a->A.nnz <= 0 || a->n_batch < 2 ||
        a->k_inner < 4 || a->k_inner >= 16)
        return false;
    const int64_t upitch = (a->patch_umax + 2) & ~1;
    const int64_t lanes = a->patch_rows > (upitch + 1) / 2
                              ? a->patch_rows
                              : (upitch + 1) / 2;
    const int64_t out_bytes =
        (int64_t)a->patch_rows * (a->k_inner * 9 + 4) + 16;
    return lanes <= 512 &&
           upitch * 4 * 16 + out_bytes <= (int64_t)kPatchLdsMax;
}

// family 11: one wave per long row x 64 columns, source cells sliding through
// LDS in windows (spmm_longwave.h).  The patch plan attached names R <= 16
// consecutive long rows per patch, row-major entries (no patch_ell_base).
bool longwave_usable(const remap_apply_args *a, const Call &c)
{
    return a->patch_ptr && a->patch_ucol && a->patch_lidx &&
           a->patch_rowptr && a->patch_val && !a->patch_ell_base &&
           a->patch_rows > 0 && a->patch_rows <= kPatchWaves &&
           a->n_patches > 0 && a->A.max_row_nnz > 0 &&
           a->patch_rows * (2 * kLongPre * kLongCellBytes +
                            ((a->A.max_row_nnz + 15) / 16 * 16 + 16) *
                                (int64_t)kLongRecordBytes) <=
               (int64_t)kPatchLdsMax &&
           c.n_rows <= a->n_patches * (int64_t)a->patch_rows &&
           c.n_rows > (a->n_patches - 1) * (int64_t)a->patch_rows;
}

// Does the LDS patch kernel (family 5) take a call of K fields?  Its time is
// flat in K up to one 64-column chunk, the lane-per-(row, k) kernel's grows
// with K: us per launch, lane-per-(row, k) (K <= 32) or scalar-cache rows /
// LDS patches, at K = 12, 16, 24, 32, 48 -- 1 deg -> 0.5 deg bilinear (2 025
// patches): 22 / 26, 28 / 27, 41 / 28, 53 / 28, 71 / 29; config 4's map (52 K
// patches): 2 807 / 2 177, 3 717 / 2 167, 5 625 / 2 316, 7 145 / 2 357,
// 9 681 / 2 806; QU240 -> 1 deg (253 patches, ONE round of workgroups): 6.4
// / 11.9, 7.4 / 11.6, 10.1 / 12.2, 12.4 / 12.5, 17.2 / 12.9.
bool patch_serves(const remap_apply_args *a, const Call &c)
{
    // (float32 fields and the masked mode cross later on the shorter list:
    // 1 deg -> 0.5 deg at K = 16, lane-per / patches: f32 24.7 / 28.9, masked
    // 27.0 / 37.0; at K = 24: 35.7 / 28.6 and 38.9 / 37.3)
    return c.patch_ok &&
           (c.K > 32 || (c.K >= 24 && a->n_patches >= 1024) ||
            (c.K >= 16 && a->n_patches >= 8192));
}

// Fewest columns the shared form of the frac_b and raw modes takes: with one
// K tile per wave it passes the 8-row groups between 100 and 112 columns
// (config 5's mapping, ms per launch, groups / shared: K = 96 2.48 / 2.54,
// 100 2.62 / 2.59, 112 2.77 / 2.65, 120 2.97 / 2.73, 128 3.12 / 2.80; 64:
// 2.29 / 2.45 -- half the lanes idle).
constexpr int64_t kShareMinK = 104;
// ... and the fewest the narrow form (one column per lane, at most 64) takes
constexpr int64_t kNarrowMinK = 34;

// The forms of family 10 that address X with a flat 64-bit address per lane
// (LDS-DMA: spmm_groupshare.h, spmm_timeshare.h, spmm_cellshare.h) also serve
// fields whose batches lie further apart than 32-bit offsets reach -- (Time,
// nCells, nVertLevels) on a 3.7 M-cell mesh: 1.9 GB per time slice.  Does
// this call take one of them?
bool wide_share(const remap_apply_args *a, const Call &c)
{
    if (!c.share_ok || !c.dma16 || a->x_src_fold != 0 ||
        a->x_row_stride < 0 || a->x_row_stride >= (int64_t(1) << 29))
        return false;
    if (a->mode == REMAP_MODE_MASKED && (a->flags & REMAP_FLAG_CELL_MASKS) &&
        a->share_waves == 4 && (a->tune[5] == 0 || a->tune[5] == 32) &&
        a->group_rows == 8 && c.K > 128)
        return true;   // spmm_cellshare.h
    if (a->mode == REMAP_MODE_MASKED)
        return a->share_waves == 4 && a->n_batch >= 3 &&
               (a->flags & (REMAP_FLAG_BATCH_MASKS | REMAP_FLAG_CELL_MASKS)) &&
               (a->tune[5] == 0 || a->tune[5] == 32);
    return a->share_waves == 4 && a->tune[5] == 32 && c.K >= kShareMinK;
}

// REMAP_FLAG_TUNE_HINT: can the preferred family serve this call?
bool hint_usable(const remap_apply_args *a, const Call &c)
{
    if (a->tune[0] == 5 && !short_runs(a))
        return patch_serves(a, c);
    // the row groups against the lane-per-(row, k) kernel at few fields
    // (tools/mid_k_probe.py; us per launch, lane-per / groups): config 3 K =
    // 24 50.6 / 52.9, 32 61.7 / 52.7; headline 24 384 / 402, 32 452 / 406;
    // config 5 (8-row groups, 12 entries per row) 16 2 157 / 1 967, 24
    // 3 102 / 2 020, 32 3 779 / 2 026
    if (a->tune[0] == 10 && !short_runs(a))
        return c.group_ok && (c.small_offsets || wide_share(a, c)) &&
               (c.K >= 28 || (c.K >= 16 && a->group_rows == 8));
    if (a->tune[0] == 7 && a->tune[2] == 2)
        return runs_usable(a, c);
    if (a->tune[0] == 7)
        return c.cell_ok;   // LDS-staged lanes across rows: any K
    if (a->tune[0] == 11)   // wave per long row, windows through LDS
        return longwave_usable(a, c);
    if (a->tune[0] == 9)    // wave per long row: any K, any layout
        return a->A.max_row_nnz > 0 &&
               2 * ((a->A.max_row_nnz + 34) * 8) <= (int64_t)kPatchLdsMax;
    if (c.K <= 32)
        return false;  // the lane-per-(row, k) kernel owns small K
    if (short_runs(a) && a->tune[0] != 4)
        return false;  // (Time, nCells): lanes across rows
    switch (a->tune[0]) {
    case 10:
        return c.group_ok && (c.small_offsets || wide_share(a, c));
    case 5:
        return c.patch_ok;
    case 7:
        return c.cell_ok;
    case 8:
        return c.strip_ok;
    case 6:
        return a->A.csr_pad >= 8 && c.small_offsets &&
               (a->tune[1] != 2 || c.can_vec2);
    default:
        return true;
    }
}

// Measured on config 3: scalar-cache metadata beats the
// plain wave-per-row kernel by ~10 %.  The LDS patch family wins when source
// rows are heavily shared (config 4) and ties otherwise: the host attaches a
// patch plan only in the first case (RemapPlan.auto_schedule), so its
// presence decides.
int automatic_family(const remap_apply_args *a, const Call &c)
{
    // few fields: lane per (row, k).  The sub-group-per-row kernel (family
    // 3, coalesced (col, S) loads, CSR-order sums by shuffles) was built for
    // this and measured slower on every case -- config 3's map: K = 1 17.7
    // vs 9.7 us, K = 12 77 vs 27 us, K = 32 268 vs 57 us; config 1's
    // bilinear map at K = 1: 10.1 vs 10.1 us -- so it stays opt-in
    if (a->x_src_fold != 0)
        return 4;
    if (!short_runs(a) && patch_serves(a, c))
        return 5;
    if (c.K <= 32)
        return 2;
    if (short_runs(a))
        return 4;
    return a->A.csr_pad >= 8 ? 6 : 1;
}

// (row blocks x K chunks) work list -> grid size, XCD-aware or not.
int shape_grid(KParams &p, int64_t n_rowblocks, int64_t n_chunks, bool xcd,
               int64_t &grid)
{
    p.n_rowblocks = n_rowblocks;
    p.n_blocks = n_rowblocks * n_chunks;
    p.xcd_map = xcd ? 1 : 0;
    p.blocks_per_xcd = (p.n_blocks + kXcds - 1) / kXcds;
    grid = xcd ? p.blocks_per_xcd * kXcds : p.n_blocks;
    if (grid <= 0 || grid > 0x7fffffffLL)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: grid of %lld blocks; split the rows",
                    (long long)grid);
    return REMAP_OK;
}

int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// K tiles for the wave-per-row(-group) kernels: `cols` columns per tile.
// Returns the number of K chunks (of `tiles` tiles each) and sets p.bpc:
// whole batches per tile when the batches are short runs that are no whole
// number of 128-byte lines -- (Time, nCells, 61 levels) -- and still fill
// >= 3/4 of a tile; else the flat column list is cut every `cols` columns
// (tile_offsets).  Measured on config 3's mapping, (Time, nCells, L) in
// place, ms per launch flat -> batch-aligned: L = 61 0.59 -> 0.43, 57 0.66 ->
// 0.54, 100 0.45 -> 0.42, 56 0.39 -> 0.36, 60 unchanged (0.43: its 480-byte
// runs straddle lines wherever the tiles are cut); L = 48 (384-byte runs,
// the flat cuts fall on line boundaries) 0.38 -> 0.40, hence the line rule.
int64_t shape_tiles(KParams &p, const remap_apply_args *a, int64_t K,
                    int cols, int tiles)
{
    p.bpc = 0;
    const int64_t ki = a->k_inner;
    if (a->n_batch > 1 && ki < cols && cols % ki != 0 && ki % 16 != 0) {
        const int64_t m = cols / ki;
        if (m * ki * 4 >= (int64_t)cols * 3) {
            p.bpc = static_cast<uint32_t>(m);
            return ceil_div(a->n_batch, m * tiles);
        }
    }
    // runs longer than one tile but not than the wave's tiles together, and
    // no whole lines: one batch per chunk if >= 3/4 of the lanes stay busy
    // (flat -> per batch: 101 levels 0.551 -> 0.522 ms, 127: 0.522 -> 0.478;
    // 81 levels 0.563 -> 0.583 and 65: 0.67 -> 0.72, hence the 3/4)
    if (a->n_batch > 1 && tiles > 1 && ki > cols &&
        ki <= (int64_t)cols * tiles && ki % 16 != 0 &&
        ki * 4 >= (int64_t)cols * tiles * 3) {
        p.bpc = kBatchPerChunk;
        return a->n_batch;
    }
    return ceil_div(K, (int64_t)cols * tiles);
}

int run_rowgroup(const remap_apply_args *a, const Call &c, KParams p,
                 hipStream_t stream)
{
    if (!c.group_ok || !(c.small_offsets || wide_share(a, c)))
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the rowgroup kernel needs the "
                    "row-group schedule for [row_begin, row_end) and 32-bit "
                    "offsets (or the shared lists and a float64 field in "
                    "whole 16-byte pieces)");
    // REMAP_FLAG_CELL_MASKS with the shared lists (remap_schedule_auto builds
    // them on entry-rich mappings): the per-row normaliser through the LDS
    // ring (spmm_cellshare.h) -- flat 64-bit addresses, so a (Time, nCells,
    // nVertLevels) field's batches are served too.  Anything it cannot serve
    // (f32, odd strides, at most 128 columns) takes the forms below;
    // tune[5] = 8 keeps the 8-row groups (spmm_groupmask.h).
    if ((a->flags & REMAP_FLAG_CELL_MASKS) && a->mode == REMAP_MODE_MASKED &&
        (a->tune[5] == 0 || a->tune[5] == 32) && a->group_rows == 8 &&
        c.share_ok &&
        a->share_waves == 4 && c.dma16 && c.K > 128 && a->x_src_fold == 0 &&
        a->x_row_stride >= 0 && a->x_row_stride < (int64_t(1) << 29)) {
        p.rows_per_wave = 1;
        const int64_t k_chunks = shape_tiles(p, a, c.K, kWave * 2, 2);
        int64_t grid;
        const int rc = shape_grid(p, ceil_div(a->n_groups, (int64_t)4),
                                  k_chunks, a->tune[4] != 1, grid);
        if (rc != REMAP_OK)
            return rc;
        if (a->tune[4] == 3)
            p.xcd_map |= 2;
        return launch_cellshare(a, p, c.fma, grid, stream);
    }
    // tune[5] = 32: the shared form (spmm_groupshare.h) -- W waves, one
    // union through an LDS ring; float64 fields in whole 16-byte pieces,
    // at least 104 columns (kShareMinK).  A call it cannot serve takes the 8-row groups
    // of the same schedule (a preference under REMAP_FLAG_TUNE_HINT, an
    // error otherwise).
    if (a->tune[5] == 32 && c.K <= kWave && c.K >= kNarrowMinK &&
        c.share_ok && a->share_waves == 4 && c.dma16 &&
        a->mode != REMAP_MODE_MASKED && a->x_src_fold == 0 &&
        a->x_row_stride >= 0 && a->x_row_stride < (int64_t(1) << 29)) {
        // at most 64 columns: a lane per column, two union entries per DMA
        // instruction (spmm_narrowshare.h)
        const int64_t k_chunks = shape_tiles(p, a, c.K, kWave, 1);
        if (p.bpc == 0) {
            p.rows_per_wave = 1;
            int64_t grid;
            const int rc = shape_grid(p, ceil_div(a->n_groups, (int64_t)4),
                                      k_chunks, a->tune[4] != 1, grid);
            if (rc != REMAP_OK)
                return rc;
            if (a->tune[4] == 3)
                p.xcd_map |= 2;
            return launch_narrowshare(a, p, c.fma, grid, stream);
        }
    }
    if (a->tune[5] == 32) {
        const bool can = c.share_ok && a->share_waves == 4 && c.dma16 &&
                         c.K >= kShareMinK &&
                         a->mode != REMAP_MODE_MASKED &&
                         a->x_src_fold == 0 && a->x_row_stride >= 0 &&
                         a->x_row_stride < (int64_t(1) << 29);
        if (can) {
            // K tiles per wave: 2 (256 columns per workgroup and step); at
            // most 128 columns -- ONE 3-D field of 104 ... 128 levels --: 1
            const int tiles = (a->tune[2] == 1 || c.K <= 128) ? 1 : 2;
            p.rows_per_wave = 1;
            const int64_t k_chunks =
                shape_tiles(p, a, c.K, kWave * 2, tiles);
            const int64_t n_super =
                ceil_div(a->n_groups, (int64_t)a->share_waves);
            int64_t grid;
            const int rc =
                shape_grid(p, n_super, k_chunks, a->tune[4] != 1, grid);
            if (rc != REMAP_OK)
                return rc;
            if (a->tune[4] == 3)
                p.xcd_map |= 2;
            return launch_groupshare(a, p, tiles, c.fma, grid, stream);
        }
        if (!(a->flags & REMAP_FLAG_TUNE_HINT))
            return fail(REMAP_ERR_UNSUPPORTED,
                        "remap_apply_f64: the shared form (tune[5] = 32) "
                        "serves the frac_b and raw modes on float64 fields "
                        "of at least 104 even-strided columns, on a plan "
                        "with share_* lists of 4 groups");
    }
    // REMAP_FLAG_BATCH_MASKS: the masked mode of a field of several batches
    // whose mask is expected not to change from batch to batch -- (Time,
    // nCells, nVertLevels) cut by bathymetry -- on 8-row groups: lanes
    // across the levels, four time slices per lane, one normaliser per lane
    // and row (spmm_grouptime.h).  A hint: a group that meets anything else
    // takes the general form inside the same launch.
    // (REMAP_FLAG_CELL_MASKS on a field whose batches lie further apart
    // than 32-bit offsets reach: land cells are missing at every time too,
    // and the form below through the LDS ring is the one that reaches)
    const bool time_form =
        (a->flags & REMAP_FLAG_BATCH_MASKS) ||
        ((a->flags & REMAP_FLAG_CELL_MASKS) && !c.small_offsets);
    if (time_form && a->mode == REMAP_MODE_MASKED &&
        a->group_rows == 8 && a->n_batch >= 3 && a->x_src_fold == 0 &&
        (a->tune[5] == 0 || a->tune[5] == 32 || a->tune[5] == 9)) {
        const int wpb = a->tune[1] == 4 ? 4 : a->tune[1] == 2 ? 2 : 1;
        p.rows_per_wave = a->tune[3] > 0 && a->tune[5] != 32 ? a->tune[3] : 1;
        p.bpc = 0;
        const int64_t n_lb = ceil_div(a->k_inner, kWave);
        const int64_t n_tb = ceil_div(a->n_batch, kTimeBlock);
        int64_t grid;
        // with the shared lists: one union per 4 x 8 tile through the LDS
        // ring (spmm_timeshare.h; tune[5] = 9 keeps the form below)
        if (c.share_ok && a->share_waves == 4 && c.dma16 &&
            a->tune[5] != 9 && a->x_row_stride >= 0 &&
            a->x_row_stride < (int64_t(1) << 29)) {
            p.rows_per_wave = 1;
            const int rc = shape_grid(p, ceil_div(a->n_groups, (int64_t)4),
                                      n_lb * n_tb, a->tune[4] != 1, grid);
            if (rc != REMAP_OK)
                return rc;
            if (a->tune[4] == 3)
                p.xcd_map |= 2;
            return launch_timeshare(a, p, c.fma, grid, stream);
        }
        if (!c.small_offsets)
            return fail(REMAP_ERR_UNSUPPORTED,
                        "remap_apply_f64: batches further apart than 32-bit "
                        "offsets reach need the shared lists");
        const int rc = shape_grid(
            p, ceil_div(a->n_groups, (int64_t)wpb * p.rows_per_wave),
            n_lb * n_tb, a->tune[4] != 1, grid);
        if (rc != REMAP_OK)
            return rc;
        if (a->tune[4] == 3)
            p.xcd_map |= 2;
        return c.f32 ? launch_grouptime<float>(a, p, wpb, c.fma, grid, stream)
                     : launch_grouptime<double>(a, p, wpb, c.fma, grid,
                                                stream);
    }
    // f32 rows are half as long: two K tiles per wave keep the bytes per
    // wave and row at 1 KiB (measured +7 % on config 3 with f32 fields).
    // f64 with 128 < K <= 224 columns: one wave over both (the second only
    // part full) beats two waves per group (config 3's map: K = 160 0.142
    // vs 0.170 ms, K = 192 0.160 vs 0.177; from K = 256 one tile wins again)
    int tiles = a->tune[2];
    if (tiles == 0)
        tiles = c.f32 ? 2 : (c.K > 128 && c.K <= 224) ? 2 : 1;
    // REMAP_FLAG_CELL_MASKS on 8-row groups: the per-row normaliser form
    // (spmm_groupmask.h) with two K tiles per wave, whatever tune[2] says
    const bool cell_masks =
        (a->flags & REMAP_FLAG_CELL_MASKS) && a->mode == REMAP_MODE_MASKED &&
        a->group_rows == 8 && c.can_vec2 && c.K > 128 &&
        (a->tune[5] == 0 || a->tune[5] == 32 || a->tune[5] == 8);
    if (cell_masks)
        tiles = 2;
    if (tiles != 2 || c.K <= 128)
        tiles = 1;
    // odd strides or level counts: one element per lane and tile, two tiles.
    // At most 64 columns (ONE 3-D field: (1, nCells, 60 levels)): one element
    // per lane fills the wave where two leave half of it idle -- config 3's
    // map, us per launch, two -> one element: masked 60 levels 83 -> 67, 64:
    // 80 -> 65, 33: 76 -> 54; frac_b 33 levels 59 -> 53, 48 / 60: a tie, 64
    // (and 2 x 32): 67 -> 70, hence "fewer than 64" outside the masked mode
    const bool narrow =
        c.K <= 64 && (a->mode == REMAP_MODE_MASKED || c.K < 64);
    const int vec = (c.can_vec2 && !narrow) ? 2 : 1;
    if (vec == 1)
        tiles = c.K > 64 ? 2 : 1;
    if (a->group_rows == 16 && vec != 2)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: 16-row groups serve float64 fields of "
                    "more than 64 even-strided columns");
    // groups per wave (a call the shared form declined: the shape the
    // entry-rich groups run in, one group per single-wave workgroup)
    int gpw = a->tune[5] == 32 ? 1 : a->tune[3] > 0 ? a->tune[3] : 2;
    // union entries in flight
    int unr = (a->tune[5] == 4 || a->tune[5] == 16) ? a->tune[5] : 8;
    // waves per workgroup (tune[1], unused otherwise by this family)
    int wpb = a->tune[5] == 32 ? 1
              : (a->tune[1] == 1 || a->tune[1] == 2) ? a->tune[1]
                                                     : kWavesPerBlock;
    // tune[5] >= 100 (diagnostic build): the lock-step experiment
    // (step-aligned lists: see spmm_rowgroup.h); the workgroup is one
    // supergroup of tune[1] groups
    const bool lock = a->tune[5] >= 100;
#ifndef REMAP_DIAG
    if (lock)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: tune[5] >= 100 is a switch of the "
                    "diagnostic build (-DREMAP_DIAG), absent from this one");
#endif
    if (lock) {
        if (c.f32 || a->group_rows != 4 || !c.can_vec2 || tiles != 1 ||
            c.fma || (a->tune[1] != 4 && a->tune[1] != 16))
            return fail(REMAP_ERR_UNSUPPORTED,
                        "remap_apply_f64: the lock-step walk serves float64, "
                        "2 x 2 groups, supergroups of 4 or 16");
        wpb = a->tune[1];
        gpw = 1;
    }
    p.rows_per_wave = gpw;
    const int64_t k_chunks = shape_tiles(p, a, c.K, kWave * vec, tiles);
    // several SHORT batches per tile -- (Time, nCells, 60 levels): two
    // 480-byte runs per source row, neither a whole number of cache lines:
    // sixteen union entries in flight instead of eight (config 3's map,
    // (8, nCells, 60), two boxes: 0.447 -> 0.426 ms, 0.412 -> 0.393; 100 or
    // 61 levels, one batch per tile: slower or no change)
    if ((a->tune[5] == 0 || a->tune[5] == 32) && p.bpc >= 2 &&
        p.bpc != kBatchPerChunk && !lock &&
        a->group_rows == 4)
        unr = 16;
    int64_t grid;
    const int rc = shape_grid(
        p, ceil_div(a->n_groups, (int64_t)wpb * gpw), k_chunks,
        a->tune[4] != 1, grid);
    if (rc != REMAP_OK)
        return rc;
    if (a->tune[4] == 3)   // the K-chunks of a row block side by side
        p.xcd_map |= 2;
    // tune[5] = 26 / 28: the rolling form, 6 / 8 union entries in flight
    const int roll = a->tune[5] == 26 ? 6 : a->tune[5] == 28 ? 8 : 0;
    return c.f32 ? launch_rowgroup<float>(a, p, tiles, unr, vec, wpb, c.fma,
                                          grid, stream, false, 0,
                                          cell_masks && vec == 2)
                 : launch_rowgroup<double>(a, p, tiles, unr, vec, wpb, c.fma,
                                           grid, stream, lock, roll,
                                           cell_masks && vec == 2);
}

int run_patch(const remap_apply_args *a, const Call &c, KParams p,
              hipStream_t stream)
{
    if (!c.patch_ok)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the patch kernel needs a patch plan "
                    "covering [row_begin, row_end)");
    // Columns per K-chunk.  At most 64 columns: 64 whatever the plan was
    // sized for (it fits a fortiori) -- every lane of the compute phase busy,
    // half the staging (config 2, K = 64: 13.7 -> 11.5 us, masked 16.0 ->
    // 12.1; config 4's map at K = 64: 4.34 -> 3.17 ms).  Odd strides or
    // level counts: 64 columns, one element per lane.  float64 rows in whole
    // 16-byte pieces go by LDS-DMA, everything else through registers.
    const int wc = (!c.can_vec2 || c.K <= 64) ? 64 : a->patch_row_bytes / 8;
    const int row_bytes = wc * 8;
    int64_t grid;
    const int rc = shape_grid(p, a->n_patches,
                              shape_tiles(p, a, c.K, wc, 1),
                              a->tune[4] == 0 || a->tune[4] == 2, grid);
    if (rc != REMAP_OK)
        return rc;
    uint32_t lds_bytes = patch_lds_bytes(a->patch_umax, a->patch_emax,
                                         a->patch_rows, row_bytes);
    if (lds_bytes < 1024)
        lds_bytes = 1024;
    // 64-column chunks (K <= 64, odd strides) on a work list of several
    // rounds: 512-thread workgroups -- four per CU instead of two, so that a
    // workgroup's gather (three dependent trips) finds three others
    // computing.  The short rows of 1 deg -> 0.5 deg bilinear (2 025
    // patches of 128 rows), us per launch 1 024 / 512 threads: K = 64 37.0 /
    // 32.4, K = 48 33.3 / 30.1, K = 32 31.7 / 27.3; a work list of ONE round
    // (QU240 -> 1 deg: 253 patches) loses: 12.2 / 15.7.  tune[1] = 512 /
    // 1024 forces one or the other.
    const bool small_blocks =
        a->tune[1] == 512 || (a->tune[1] != 1024 && grid >= 4 * 256);
    const int block = (wc == 64 && small_blocks && a->patch_rows <= 512)
                          ? 512
                          : kPatchBlock;
    patch_fn pf = pick_patch(c.f32, a->mode, c.fma, wc, c.dma16, block);
    REMAP_HIP_CHECK(diag_lds_throttle(a, reinterpret_cast<const void *>(pf),
                                      lds_bytes));
    if (lds_bytes > 64 * 1024)
        REMAP_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(pf),
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(pf, dim3(static_cast<uint32_t>(grid)),
                       dim3(block), lds_bytes, stream, p, a->flags,
                       a->patch_rowptr, a->patch_val, a->patch_lidx,
                       a->patch_ptr, a->patch_ucol, a->row_order, a->frac_b,
                       a->patch_rows, a->patch_umax, a->patch_emax,
                       a->n_patches);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

int launch_plain(kernel_fn fn, const remap_apply_args *a, const KParams &p,
                 int64_t grid, hipStream_t stream)
{
    uint32_t lds_bytes = 0;
    REMAP_HIP_CHECK(diag_lds_throttle(a, reinterpret_cast<const void *>(fn),
                                      lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)), dim3(kBlock),
                       lds_bytes, stream, p, a->flags);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

int run_rowlane(const remap_apply_args *a, const Call &c, const KParams &p,
                hipStream_t stream)
{
    const int64_t grid = ceil_div(c.n_rows * c.K, kBlock);
    if (grid <= 0 || grid > 0x7fffffffLL)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: grid of %lld blocks; split the rows",
                    (long long)grid);
    int unr = a->tune[1];
    if (unr != 1 && unr != 4 && unr != 8)
        unr = 4;
    return launch_plain(c.f32 ? pick_rowlane<float>(unr, a->mode, c.fma)
                              : pick_rowlane<double>(unr, a->mode, c.fma),
                        a, p, grid, stream);
}

// family 4: lanes across rows, TT fields per lane
int run_rowcell(const remap_apply_args *a, const Call &c, KParams p,
                hipStream_t stream)
{
    int tt = a->tune[1];
    if (tt != 4 && tt != 8 && tt != 16)
        tt = 8;
    int unr = a->tune[2];
    if (unr != 1 && unr != 2 && unr != 4)
        unr = 2;
    p.row_order = nullptr;   // lanes = consecutive rows: coalesced Y stores
    int64_t grid;
    const int rc = shape_grid(p, ceil_div(c.n_rows, kBlock),
                              ceil_div(c.K, tt), a->tune[4] != 1, grid);
    if (rc != REMAP_OK)
        return rc;
    return launch_plain(c.f32 ? pick_rowcell<float>(tt, unr, a->mode, c.fma)
                              : pick_rowcell<double>(tt, unr, a->mode, c.fma),
                        a, p, grid, stream);
}

// family 7: LDS-staged patches, lanes across rows, TT fields per lane.
// tune[1] = TT (4, 8, 16; 0: the largest whose LDS image stays under 32 KB,
// so that several workgroups share a CU and overlap staging with compute)
int run_patchcell(const remap_apply_args *a, const Call &c, KParams p,
                  hipStream_t stream)
{
    if (!c.cell_ok)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the patchcell kernel needs a patch "
                    "plan covering [row_begin, row_end)");
    const int32_t upitch = (a->patch_umax + 2) & ~1;
    int tt = a->tune[1];
    const bool long_rows = a->patch_ell_base != nullptr;
    if (long_rows && (tt == 1 || tt == 2)) {
        // (column-major plans only)
    } else if (tt != 4 && tt != 8 && tt != 16) {
        tt = 16;
        while (tt > 4 && (int64_t)upitch * tt * 8 > 32 * 1024)
            tt >>= 1;
    }
    while (tt > 4 && (int64_t)upitch * tt * 8 > (int64_t)kPatchLdsMax)
        tt >>= 1;
    // Row-major plans whose patches fit one lane per row and two cells per
    // lane: the workgroup stays on its patch over a run of chunks
    // (spmm_patchtime; tune[2] = 1 keeps the one-chunk kernel).  Runs are
    // sized so that ~2 000 workgroups exist: (120, nCells) on 1 020 patches:
    // 2 runs of 8 chunks.
    if (!long_rows && a->tune[1] == 2)
        tt = 2;     // (the persistent kernel only)
    // one lane per row and at most two cells per lane: 256 threads for the
    // 16 x 16 patches, 512 / 1024 for larger ones
    int64_t lanes = a->patch_rows > (upitch + 1) / 2 ? a->patch_rows
                                                      : (upitch + 1) / 2;
    const int block = lanes <= 256 ? 256 : lanes <= 512 ? 512 : 1024;
    const bool persistent =
        !long_rows && a->tune[2] != 1 && a->A.nnz > 0 &&
        (tt == 2 || tt == 4 || tt == 8 || tt == 16) && lanes <= 1024 &&
        (int64_t)upitch * 2 * 16 <= (int64_t)kPatchLdsMax;
    if (!persistent && !long_rows && tt == 2)
        tt = 4;   // (the one-chunk kernel stages 4, 8 or 16 fields)
    if (persistent && tt == 16)
        tt = 8;
    // (two LDS images: fewer fields per lane until they fit)
    while (persistent && tt > 2 &&
           (int64_t)upitch * tt * 16 > (int64_t)kPatchLdsMax)
        tt >>= 1;
    // tune[2] = 2: fields with short runs in several batches, (Time, nCells,
    // 4 ... 15) -- a batch at a time, results written out through LDS
    // (spmm_patchtime<..., RUNS>)
    const int64_t ki = a->k_inner;
    const int64_t out_bytes = (int64_t)a->patch_rows * (ki * 9 + 4) + 16;
    const bool runs = persistent && a->tune[2] == 2 && runs_usable(a, c);
    if (a->tune[2] == 2 && !runs)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: tune[2] = 2 (short runs, a batch at a "
                    "time) needs 4 <= k_inner < 16, several batches and a "
                    "patch plan of at most 512 rows / 1 022 cells per patch");
    // columns per chunk of the batch-at-a-time kernel: 4, or tune[1] = 6
    // where the LDS image(s) still fit (a batch that is ONE chunk needs one
    // image, else two)
    if (runs) {
        const int want = a->tune[1];
        tt = 4;
        if (want == 6) {
            const int64_t images = ki <= want ? 1 : 2;
            if ((int64_t)upitch * want * 8 * images + out_bytes <=
                (int64_t)kPatchLdsMax)
                tt = want;
        }
    }
    const bool one_image = runs && ki <= tt;
    const int64_t sub = runs ? ceil_div(ki, tt) : 1;
    const int64_t n_chunks = runs ? a->n_batch * sub : ceil_div(c.K, tt);
    int64_t groups = 1, cpw = 1;
    if (persistent) {
        // Runs of chunks per patch: ONE round of workgroups on the chip --
        // as many as fit at a time (threads, LDS) -- each walking all its
        // chunks.  Config 3's map, (120, nCells) cold, 4 fields per lane:
        // 32 x 32 patches (254 workgroups of 1 024 threads, one per CU) 1 /
        // 4 runs 0.121 / 0.149 ms; 16 x 16 patches (1 013 of 256 threads, 8
        // per CU) 1 / 2 / 4 runs 0.146 / 0.141 / 0.144.
        const int64_t lds_wg =
            (int64_t)upitch * tt * (one_image ? 8 : 16) +
            (runs ? out_bytes : 0);
        int64_t fit = 2048 / block;
        if (fit > (int64_t)kPatchLdsMax / lds_wg)
            fit = (int64_t)kPatchLdsMax / lds_wg;
        if (fit > 8)
            fit = 8;
        if (fit < 1)
            fit = 1;
        groups = (256 * fit) / a->n_patches;
        if (groups < 1)
            groups = 1;
        if (a->tune[3] > 0)
            groups = a->tune[3];
        if (groups > n_chunks / sub)
            groups = n_chunks / sub;
        // (RUNS: whole batches per workgroup)
        cpw = ceil_div(ceil_div(n_chunks, groups), sub) * sub;
        groups = ceil_div(n_chunks, cpw);
        p.rows_per_wave = static_cast<int32_t>(cpw);
    }
    // short level runs of even length on even strides from a 16-byte (f32:
    // 8-byte) aligned base: two elements per load (tune[5] = 1: one)
    {
        const size_t pair = (c.f32 ? sizeof(float) : sizeof(double)) * 2;
        p.x_pairs = runs && ki % 2 == 0 && a->x_row_stride % 2 == 0 &&
                    a->x_batch_stride % 2 == 0 && a->x_src_fold == 0 &&
                    aligned(a->X, pair) && a->tune[5] != 1;
        p.y_pairs = runs && ki % 2 == 0 && a->y_row_stride % 2 == 0 &&
                    a->y_batch_stride % 2 == 0 && aligned(a->Y, 16) &&
                    a->tune[5] != 1;
    }
    int64_t grid;
    const int rc = shape_grid(p, a->n_patches,
                              persistent ? groups : n_chunks,
                              a->tune[4] != 1, grid);
    if (rc != REMAP_OK)
        return rc;
    uint32_t lds_bytes =
        static_cast<uint32_t>(upitch) * tt * 8u *
            (persistent && !one_image ? 2u : 1u) +
        (runs ? static_cast<uint32_t>(out_bytes) : 0u);
    if (lds_bytes < 1024)
        lds_bytes = 1024;
    const int layout = a->patch_ell_base ? 1 : 0;
    const int launch_block = persistent ? block : kCellBlock;
    cell_fn fn =
        runs ? (c.f32 ? pick_patchruns<float>(a->mode, c.fma, block, tt)
                      : pick_patchruns<double>(a->mode, c.fma, block, tt))
        : persistent
            ? (c.f32 ? pick_patchtime<float>(tt, a->mode, c.fma, block)
                     : pick_patchtime<double>(tt, a->mode, c.fma, block))
            : (c.f32 ? pick_patchcell<float>(tt, a->mode, c.fma, layout)
                     : pick_patchcell<double>(tt, a->mode, c.fma, layout));
    if (lds_bytes > 64 * 1024)
        REMAP_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(fn),
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)),
                       dim3(launch_block), lds_bytes, stream, p, a->flags,
                       a->patch_rowptr,
                       a->patch_val, a->patch_lidx,
                       a->patch_ptr, a->patch_ucol, a->row_order, a->frac_b,
                       a->patch_rows, upitch, a->n_patches,
                       a->patch_ell_base);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

typedef void (*strip_fn)(const KParams, const int32_t *, const int32_t *,
                         const int32_t *, const int32_t *, const int64_t *,
                         const char *, int32_t, int32_t, int32_t, int32_t,
                         int32_t, int64_t);

template <int DEPTH>
strip_fn pick_strip_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_strip<REMAP_MODE_RAW, true, DEPTH>
                   : spmm_strip<REMAP_MODE_RAW, false, DEPTH>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_strip<REMAP_MODE_FRACB, true, DEPTH>
                   : spmm_strip<REMAP_MODE_FRACB, false, DEPTH>;
    default:
        return fma ? spmm_strip<REMAP_MODE_MASKED, true, DEPTH>
                   : spmm_strip<REMAP_MODE_MASKED, false, DEPTH>;
    }
}

// family 8: one workgroup per (strip segment, 64-column K-chunk), unit-major
int run_strip(const remap_apply_args *a, const Call &c, KParams p,
              hipStream_t stream)
{
    if (!c.strip_ok)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the strip kernel needs a strip "
                    "schedule, the whole row range and one batch of "
                    "contiguous float64 columns");
    const remap_strips *st = a->strips;
    const int64_t n_chunks = ceil_div(c.K, kStripRowBytes / 8);
    // unit-major work list: shape_grid's "chunks" are the slow index
    int64_t grid;
    const int rc = shape_grid(p, n_chunks, st->n_units, a->tune[4] != 1,
                              grid);
    if (rc != REMAP_OK)
        return rc;
    strip_fn fn;
    switch (st->depth) {
    case 1: fn = pick_strip_mode<1>(a->mode, c.fma); break;
    case 2: fn = pick_strip_mode<2>(a->mode, c.fma); break;
    case 3: fn = pick_strip_mode<3>(a->mode, c.fma); break;
    case 4: fn = pick_strip_mode<4>(a->mode, c.fma); break;
    case 5: fn = pick_strip_mode<5>(a->mode, c.fma); break;
    default: fn = pick_strip_mode<6>(a->mode, c.fma); break;
    }
    // + 256 bytes of slack: the compute waves read a row's records sixteen
    // at a time (lane l: record l % 16) whatever the row holds -- on the
    // last row of a block that fills its meta slot the look-ahead read runs
    // up to 192 bytes past the block; the values are never used, the
    // addresses stay inside the allocation
    const uint32_t lds_bytes =
        static_cast<uint32_t>(st->ring_slots + 2) * kStripRowBytes +
        static_cast<uint32_t>(st->depth + 1) * st->meta_slot_bytes + 256;
    if (lds_bytes > 64 * 1024)
        REMAP_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(fn),
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)),
                       dim3((st->waves + st->depth) * kWave), lds_bytes,
                       stream, p, st->unit_steps, st->arr_ptr, st->arr_src,
                       st->arr_slot, st->meta_ptr,
                       static_cast<const char *>(st->meta),
                       st->steps_per_unit, st->rows_per_wave, st->ring_slots,
                       st->meta_slot_bytes, st->waves, n_chunks);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

typedef void (*long_fn)(const KParams, const uint32_t, const int32_t);

template <typename XT, int TT>
long_fn pick_longrow_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_longrow<XT, REMAP_MODE_RAW, true, TT>
                   : spmm_longrow<XT, REMAP_MODE_RAW, false, TT>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_longrow<XT, REMAP_MODE_FRACB, true, TT>
                   : spmm_longrow<XT, REMAP_MODE_FRACB, false, TT>;
    default:
        return fma ? spmm_longrow<XT, REMAP_MODE_MASKED, true, TT>
                   : spmm_longrow<XT, REMAP_MODE_MASKED, false, TT>;
    }
}

template <typename XT>
long_fn pick_longrow(int tt, int mode, bool fma)
{
    switch (tt) {
    case 1: return pick_longrow_mode<XT, 1>(mode, fma);
    case 2: return pick_longrow_mode<XT, 2>(mode, fma);
    case 4: return pick_longrow_mode<XT, 4>(mode, fma);
    case 8: return pick_longrow_mode<XT, 8>(mode, fma);
    default: return pick_longrow_mode<XT, 16>(mode, fma);
    }
}

// family 9: one wave per (long row, TT flat columns).  tune[1] = TT (1, 2,
// 4, 8, 16; 0: the largest that covers K and keeps the wave's LDS image
// under 16 KB, so that ten or more waves share a CU)
int run_longrow(const remap_apply_args *a, const Call &c, KParams p,
                hipStream_t stream)
{
    if (a->A.max_row_nnz <= 0)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the long-row kernel needs "
                    "A.max_row_nnz");
    // pitch = 2 (mod 32) doubles: columns start 16-byte aligned and the lanes
    // of the sum phase, one column each, hit different LDS banks
    const int64_t pitch = (a->A.max_row_nnz + 29) / 32 * 32 + 2;
    const bool masked = a->mode == REMAP_MODE_MASKED;
    auto arrays = [&](int t) {
        return c.fma ? t + 1 : masked ? 2 * t : t;
    };
    int tt = a->tune[1];
    if (tt != 1 && tt != 2 && tt != 4 && tt != 8 && tt != 16) {
        tt = 16;
        while (tt > 1 &&
               (tt / 2 >= c.K || arrays(tt) * pitch * 8 > 16 * 1024))
            tt >>= 1;
    }
    // (an asked-for width that does not fit gives way to one that does)
    while (tt > 1 && arrays(tt) * pitch * 8 > (int64_t)kPatchLdsMax)
        tt >>= 1;
    const int64_t lds = arrays(tt) * pitch * 8;
    if (lds > (int64_t)kPatchLdsMax)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: rows of %lld entries do not fit the "
                    "long-row kernel's LDS image", (long long)a->A.max_row_nnz);
    int64_t grid;
    const int rc = shape_grid(p, c.n_rows, ceil_div(c.K, tt),
                              a->tune[4] != 1, grid);
    if (rc != REMAP_OK)
        return rc;
    const long_fn fn = c.f32 ? pick_longrow<float>(tt, a->mode, c.fma)
                             : pick_longrow<double>(tt, a->mode, c.fma);
    if (lds > 64 * 1024)
        REMAP_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(fn),
            hipFuncAttributeMaxDynamicSharedMemorySize,
            static_cast<int>(lds)));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)), dim3(kWave),
                       static_cast<uint32_t>(lds), stream, p, a->flags,
                       static_cast<int32_t>(pitch));
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

template <typename XT>
patch_fn pick_longwave(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_longwave<XT, REMAP_MODE_RAW, true>
                   : spmm_longwave<XT, REMAP_MODE_RAW, false>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_longwave<XT, REMAP_MODE_FRACB, true>
                   : spmm_longwave<XT, REMAP_MODE_FRACB, false>;
    default:
        return fma ? spmm_longwave<XT, REMAP_MODE_MASKED, true>
                   : spmm_longwave<XT, REMAP_MODE_MASKED, false>;
    }
}

int run_longwave(const remap_apply_args *a, const Call &c, KParams p,
                 hipStream_t stream)
{
    if (!longwave_usable(a, c))
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the long-wave kernel needs a row-major "
                    "patch plan of at most %d rows per patch covering "
                    "[row_begin, row_end)", kPatchWaves);
    int64_t grid;
    const int rc = shape_grid(p, a->n_patches,
                              shape_tiles(p, a, c.K, kWave, 1),
                              a->tune[4] == 2, grid);
    if (rc != REMAP_OK)
        return rc;
    // per wave: two windows of 8 cells, 512 bytes per cell, and the row's
    // records (a multiple of 16, + 16: the last batch is read whole)
    const int64_t epitch = (a->A.max_row_nnz + 15) / 16 * 16 + 16;
    const int64_t lds_need =
        a->patch_rows *
        (2 * kLongPre * kLongCellBytes + epitch * kLongRecordBytes);
    if (a->A.max_row_nnz <= 0 || lds_need > (int64_t)kPatchLdsMax)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: %d rows of up to %lld entries do not "
                    "fit the long-wave kernel's LDS image (A.max_row_nnz "
                    "must be given)", a->patch_rows,
                    (long long)a->A.max_row_nnz);
    const uint32_t lds_bytes = static_cast<uint32_t>(lds_need);
    const patch_fn fn = c.f32 ? pick_longwave<float>(a->mode, c.fma)
                              : pick_longwave<double>(a->mode, c.fma);
    if (lds_bytes > 64 * 1024)
        REMAP_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(fn),
            hipFuncAttributeMaxDynamicSharedMemorySize,
            static_cast<int>(lds_bytes)));
#ifdef REMAP_DIAG
    // (experiment, diagnostic build: the long rows' launch without the AQL
    // barrier bit -- hipExtAnyOrderLaunch -- so that it may overlap the short
    // rows' launch in front of it; hip_ext.h says gfx9 ignores the flag)
    if (getenv("REMAP_ANY_ORDER")) {
        hipExtLaunchKernelGGL(
            fn, dim3(static_cast<uint32_t>(grid)),
            dim3(static_cast<uint32_t>(a->patch_rows) * kWave), lds_bytes,
            stream, nullptr, nullptr, hipExtAnyOrderLaunch, p, a->flags,
            a->patch_rowptr, a->patch_val, a->patch_lidx, a->patch_ptr,
            a->patch_ucol, a->row_order, a->frac_b, a->patch_rows,
            a->patch_umax, static_cast<int32_t>(epitch), a->n_patches);
        REMAP_HIP_CHECK(hipGetLastError());
        return REMAP_OK;
    }
#endif
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)),
                       dim3(static_cast<uint32_t>(a->patch_rows) * kWave),
                       lds_bytes, stream, p, a->flags, a->patch_rowptr,
                       a->patch_val, a->patch_lidx, a->patch_ptr,
                       a->patch_ucol, a->row_order, a->frac_b, a->patch_rows,
                       a->patch_umax, static_cast<int32_t>(epitch),
                       a->n_patches);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// family 3: a sub-group of 8 (4 for rows of at most 4 entries: bilinear
// maps) lanes per row; tune[1] overrides the sub-group size
int run_rowsub(const remap_apply_args *a, const Call &c, const KParams &p,
               hipStream_t stream)
{
    int sub = a->tune[1];
    if (sub == 0)
        sub = (a->A.max_row_nnz > 0 && a->A.max_row_nnz <= 4) ? 4 : 8;
    if (sub != 4 && sub != 8)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[1] = %d", sub);
    const int64_t grid = ceil_div(c.n_rows, kBlock / sub);
    if (grid <= 0 || grid > 0x7fffffffLL)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: grid of %lld blocks; split the rows",
                    (long long)grid);
    const bool tree = (a->flags & REMAP_FLAG_TREE) != 0;
    return launch_plain(
        c.f32 ? pick_rowsub<float>(sub, tree, a->mode, c.fma)
              : pick_rowsub<double>(sub, tree, a->mode, c.fma),
        a, p, grid, stream);
}

// families 1 (vector-memory metadata) and 6 (scalar-cache metadata)
int run_rowwave(const remap_apply_args *a, const Call &c, KParams p,
                int family, hipStream_t stream)
{
    int vec = a->tune[1];
    if (vec == 0)
        vec = (c.can_vec2 && c.K > 64) ? 2 : 1;
    if (vec == 2 && !c.can_vec2)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: 2 elements per lane need even "
                    "strides and 16-byte aligned X/Y");
    if (vec != 1 && vec != 2)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[1] = %d", vec);
    int tiles = a->tune[2];
    if (tiles == 0)
        tiles = ((family == 1 || c.f32) && c.K >= 256) ? 2 : 1;  // measured
    if (vec == 1) {
        // odd strides / level counts: two 64-column tiles keep a wave over
        // 128 columns (rowscalar only; tune[2] = 1 forces one)
        tiles = (family == 6 && c.K > 64 && a->tune[2] != 1) ? 2 : 1;
    }
    if (tiles != 1 && tiles != 2 && tiles != 4)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[2] = %d", tiles);
    if (family == 6 && tiles == 4)
        tiles = 2;
    if (family == 6 && a->A.csr_pad < 8)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the rowscalar kernels need "
                    "csr_pad >= 8 readable entries behind col/val");
    if (family == 6 && !c.small_offsets) {
        if (a->tune[0] == 6)
            return fail(REMAP_ERR_UNSUPPORTED,
                        "remap_apply_f64: batch stride beyond the "
                        "32-bit offsets of the rowscalar kernels");
        family = 1;
    }
    const int rpw = a->tune[3] == 0 ? 4 : a->tune[3];
    if (rpw < 1 || rpw > 1024)
        return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[3] = %d", rpw);
    p.rows_per_wave = rpw;
    int64_t grid;
    const int rc = shape_grid(
        p, ceil_div(c.n_rows, (int64_t)kWavesPerBlock * rpw),
        shape_tiles(p, a, c.K, kWave * vec, tiles),
        a->tune[4] == 0 || a->tune[4] == 2, grid);
    if (rc != REMAP_OK)
        return rc;
    if (family == 6)
        return c.f32 ? launch_rowscalar<float>(a, p, vec, tiles, c.fma, grid,
                                               stream)
                     : launch_rowscalar<double>(a, p, vec, tiles, c.fma,
                                                grid, stream);
    return launch_plain(
        c.f32 ? pick_rowwave_shape<float>(vec, tiles, a->mode, c.fma)
              : pick_rowwave_shape<double>(vec, tiles, a->mode, c.fma),
        a, p, grid, stream);
}

}  // namespace

int apply(const remap_apply_args *a, hipStream_t stream)
{
    Call c;
    const int rc = check_args(a, c);
    if (rc != REMAP_OK || c.K == 0)
        return rc;
    // a preferred family that cannot serve this call gives way to the
    // automatic choice instead of failing
    remap_apply_args relaxed;
    if ((a->flags & REMAP_FLAG_TUNE_HINT) && a->tune[0] != 0 &&
        !hint_usable(a, c)) {
        relaxed = *a;
        for (int t = 0; t < 8; ++t)
            relaxed.tune[t] = 0;
        a = &relaxed;
    }
    const int family = a->tune[0] != 0 ? a->tune[0] : automatic_family(a, c);
    const KParams p = base_params(a, c);
    switch (family) {
    case 10:
        return run_rowgroup(a, c, p, stream);
    case 5:
        return run_patch(a, c, p, stream);
    case 2:
        return run_rowlane(a, c, p, stream);
    case 3:
        return run_rowsub(a, c, p, stream);
    case 4:
        return run_rowcell(a, c, p, stream);
    case 7:
        return run_patchcell(a, c, p, stream);
    case 8:
        return run_strip(a, c, p, stream);
    case 9:
        return run_longrow(a, c, p, stream);
    case 11:
        return run_longwave(a, c, p, stream);
    case 1:
    case 6:
        return run_rowwave(a, c, p, family, stream);
    default:
        return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[0] = %d", family);
    }
}

// ---------------------------------------------------------------------------
// streaming copy: the box's achievable HBM ceiling
// ---------------------------------------------------------------------------
namespace {
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void stream_copy_kernel(
    u4 *__restrict__ dst, const u4 *__restrict__ src, size_t n16)
{
    // one 16-byte element per lane, blocks walk the buffer in dispatch order
    // (measured faster on MI355X than a grid-stride loop: 6.2-6.5 vs
    // 4.6-5.7 TB/s, tools/hbm_ceiling.hip)
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n16)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i),
                                    dst + i);
}
}  // namespace

// ---------------------------------------------------------------------------
// NaN scan: the device-side half of remap_numpy.py:201-204 (`isnan(values)
// .any()` decides between the masked and the unmasked branch)
// ---------------------------------------------------------------------------
namespace {
template <typename T, bool KINDS>
__global__ __launch_bounds__(kBlock) void scan_nan_kernel(
    const T *__restrict__ x0, size_t n0, size_t head,
    int32_t *__restrict__ flag)
{
    // 16 bytes per lane per step, blocks in dispatch order; the `head`
    // elements in front of the first 16-byte boundary (a view that starts
    // inside an allocation: big[1:]) and the tail are peeled by block 0
    constexpr int PER = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(PER)));
    const T *__restrict__ x = x0 + head;
    const size_t n = n0 - head;
    const size_t nvec = n / PER;
    bool found = false;
    bool mixed = false;   // KINDS: a wave's run holds NaNs and numbers
    if (blockIdx.x == 0 && threadIdx.x < head) {
        const T t = x0[threadIdx.x];
        found |= (t != t);
        if constexpr (KINDS)
            mixed = true;   // (peeled elements: no run to judge them by)
    }
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * kBlock) {
        const vec_t v = __builtin_nontemporal_load(
            reinterpret_cast<const vec_t *>(x) + i);
        bool some = false, every = true;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            some |= (v[e] != v[e]);
            every &= (v[e] != v[e]);
        }
        found |= some;
        if constexpr (KINDS) {
            // the lanes of this wave hold one aligned run of 64 x 16 bytes
            // (fewer at the buffer's end)
            if (__ballot(some) != 0 && __ballot(every) != __ballot(true))
                mixed = true;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * PER) {
        const T t = x[nvec * PER + threadIdx.x];
        found |= (t != t);
        if constexpr (KINDS)
            mixed = true;
    }
    if constexpr (KINDS) {
        const bool any = __any(found);
        const bool mix = __any(mixed);
        if (any && (threadIdx.x & (kWave - 1)) == 0) {
            atomicOr(flag, 1);
            atomicOr(flag + 1, mix ? 3 : 1);
        }
    } else {
        if (__any(found) && (threadIdx.x & (kWave - 1)) == 0)
            atomicOr(flag, 1);
    }
}
}  // namespace

int scan_nan(const void *x, int32_t dtype, int64_t n, int32_t *flag,
             hipStream_t stream, bool kinds = false)
{
    if (n < 0 || !flag || (n > 0 && !x))
        return fail(REMAP_ERR_ARG, "remap_scan_nan: bad argument");
    if (dtype != REMAP_DTYPE_F64 && dtype != REMAP_DTYPE_F32)
        return fail(REMAP_ERR_ARG, "remap_scan_nan: unknown dtype %d", dtype);
    if (n == 0)
        return REMAP_OK;
    const size_t elem = dtype == REMAP_DTYPE_F64 ? 8 : 4;
    if (!aligned(x, elem))
        return fail(REMAP_ERR_ARG,
                    "remap_scan_nan: the buffer is not element-aligned");
    const size_t mis = reinterpret_cast<uintptr_t>(x) % 16;
    size_t head = mis ? (16 - mis) / elem : 0;
    if (head > (size_t)n)
        head = (size_t)n;
    const size_t nvec = ((size_t)n - head) * elem / 16;
    size_t grid = (nvec + kBlock - 1) / kBlock;
    if (grid < 1)
        grid = 1;
    if (grid > 256 * 64)
        grid = 256 * 64;   // grid-stride beyond 64 blocks per CU
    if (dtype == REMAP_DTYPE_F64) {
        void (*fn)(const double *, size_t, size_t, int32_t *) =
            scan_nan_kernel<double, false>;
        if (kinds)
            fn = scan_nan_kernel<double, true>;
        hipLaunchKernelGGL(fn, dim3((uint32_t)grid), dim3(kBlock), 0, stream,
                           static_cast<const double *>(x), (size_t)n, head,
                           flag);
    } else {
        void (*fn)(const float *, size_t, size_t, int32_t *) =
            scan_nan_kernel<float, false>;
        if (kinds)
            fn = scan_nan_kernel<float, true>;
        hipLaunchKernelGGL(fn, dim3((uint32_t)grid), dim3(kBlock), 0, stream,
                           static_cast<const float *>(x), (size_t)n, head,
                           flag);
    }
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// ---------------------------------------------------------------------------
// The same scan, aware of the field's LAYOUT (round 6): what is missing is
// judged per source cell and per batch, where the cells and batches are --
// not in aligned runs of the flat buffer, which a (Time, nCells, nVertLevels)
// field with land cells reads as "column by column".  One wave per source
// cell, lanes across its k_inner contiguous values, batch after batch.
// ---------------------------------------------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(kBlock) void scan_layout_kernel(
    const T *__restrict__ x, int64_t n_rows, int64_t n_batch, int64_t k_inner,
    int64_t rs, int64_t bs, int32_t *__restrict__ kinds)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave0 =
        ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave);
    bool found = false;   // a NaN
    bool part = false;    // a cell missing in some of its columns only
    bool vary = false;    // a (cell, k) missing in some batches only
    for (int64_t a = wave0; a < n_rows; a += n_waves) {
        bool any_c = false, all_c = true;
        const T *__restrict__ xa = x + a * rs;
        for (int64_t k0 = 0; k0 < k_inner; k0 += kWave) {
            const int64_t k = k0 + lane;
            const bool on = k < k_inner;
            const T *__restrict__ xk = xa + (on ? k : 0);
            bool first = false;
            int64_t b = 0;
            for (; b + 4 <= n_batch; b += 4) {   // four loads in flight
                T v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] = __builtin_nontemporal_load(xk + (b + q) * bs);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool m = on && (v[q] != v[q]);
                    if (b + q == 0)
                        first = m;
                    vary |= (m != first);
                    any_c |= m;
                    all_c &= (m || !on);
                }
            }
            for (; b < n_batch; ++b) {
                const T t = __builtin_nontemporal_load(xk + b * bs);
                const bool m = on && (t != t);
                if (b == 0)
                    first = m;
                vary |= (m != first);
                any_c |= m;
                all_c &= (m || !on);
            }
        }
        if (__any(any_c)) {
            found = true;
            if (__any(!all_c))
                part = true;
        }
    }
    const bool v = __any(vary);
    if (found && lane == 0) {
        atomicOr(kinds, 1);
        atomicOr(kinds + 1, part ? 3 : 1);
        atomicOr(kinds + 2, v ? 3 : 1);
    }
}

// kinds[3]: which launch the field gets (0 no NaN; 1 whole cells missing; 2
// the same mask in every batch; 3 anything else)
__global__ void scan_layout_form(int32_t *__restrict__ kinds, int multi)
{
    kinds[3] = kinds[0] == 0   ? 0
               : kinds[1] == 1 ? 1
               : (multi && kinds[2] == 1) ? 2
                                          : 3;
}
}  // namespace

int scan_nan_layout(const void *x, int32_t dtype, int64_t n_rows,
                    int64_t n_batch, int64_t k_inner, int64_t rs, int64_t bs,
                    int32_t *kinds, hipStream_t stream)
{
    if (n_rows < 0 || n_batch < 0 || k_inner < 0 || rs < 0 || bs < 0 ||
        !kinds || (n_rows > 0 && n_batch > 0 && k_inner > 0 && !x))
        return fail(REMAP_ERR_ARG, "remap_scan_nan_layout: bad argument");
    if (dtype != REMAP_DTYPE_F64 && dtype != REMAP_DTYPE_F32)
        return fail(REMAP_ERR_ARG, "remap_scan_nan_layout: unknown dtype %d",
                    dtype);
    if (x && !aligned(x, dtype == REMAP_DTYPE_F64 ? 8 : 4))
        return fail(REMAP_ERR_ARG,
                    "remap_scan_nan_layout: the buffer is not "
                    "element-aligned");
    if (n_rows > 0 && n_batch > 0 && k_inner > 0) {
        int64_t grid = (n_rows + kBlock / kWave - 1) / (kBlock / kWave);
        if (grid > 256 * 64)
            grid = 256 * 64;
        if (dtype == REMAP_DTYPE_F64)
            hipLaunchKernelGGL(scan_layout_kernel<double>,
                               dim3((uint32_t)grid), dim3(kBlock), 0, stream,
                               static_cast<const double *>(x), n_rows,
                               n_batch, k_inner, rs, bs, kinds);
        else
            hipLaunchKernelGGL(scan_layout_kernel<float>,
                               dim3((uint32_t)grid), dim3(kBlock), 0, stream,
                               static_cast<const float *>(x), n_rows, n_batch,
                               k_inner, rs, bs, kinds);
        REMAP_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(scan_layout_form, dim3(1), dim3(1), 0, stream, kinds,
                       n_batch >= 3 ? 1 : 0);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// one wave: shader cycles and 100 MHz ticks across a spin of `ticks` ticks
__global__ __launch_bounds__(kWave) void clock_probe_kernel(
    long long *__restrict__ out, long long ticks)
{
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(4);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
}

int clock_probe(int64_t *ticks_out, int32_t micros, hipStream_t stream)
{
    if (!ticks_out || micros < 1 || micros > 1000)
        return fail(REMAP_ERR_ARG,
                    "remap_clock_probe: needs an output and 1 ... 1000 us");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(kWave), 0, stream,
                       reinterpret_cast<long long *>(ticks_out),
                       static_cast<long long>(micros) * 100);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

int stream_copy(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0)
        return REMAP_OK;
    if (!dst || !src || bytes % 16 != 0 || !aligned(dst, 16) ||
        !aligned(src, 16))
        return fail(REMAP_ERR_ARG,
                    "remap_stream_copy: needs 16-byte aligned buffers and a "
                    "multiple of 16 bytes");
    const size_t n16 = bytes / 16;
    const size_t grid = (n16 + kBlock - 1) / kBlock;
    if (grid > 0x7fffffffull)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_stream_copy: more than 2^31 blocks");
    hipLaunchKernelGGL(stream_copy_kernel, dim3((uint32_t)grid), dim3(kBlock),
                       0, stream, static_cast<u4 *>(dst),
                       static_cast<const u4 *>(src), n16);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

}  // namespace remap

extern "C" {

int remap_abi_version(void) { return REMAP_ABI_VERSION; }

const char *remap_arch(void) { return "gfx950"; }

const char *remap_last_error(void) { return remap::error_buffer(); }

int remap_device_count(void)
{
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return remap::hip_fail(err, "hipGetDeviceCount");
    }
    return n;
}

int remap_apply_f64(const remap_apply_args *args, void *stream)
{
    return remap::apply(args, static_cast<hipStream_t>(stream));
}

int remap_scan_nan(const void *x, int32_t x_dtype, int64_t n, int32_t *flag,
                   void *stream)
{
    return remap::scan_nan(x, x_dtype, n, flag,
                           static_cast<hipStream_t>(stream));
}

int remap_scan_nan_kinds(const void *x, int32_t x_dtype, int64_t n,
                         int32_t *kinds, void *stream)
{
    return remap::scan_nan(x, x_dtype, n, kinds,
                           static_cast<hipStream_t>(stream), true);
}

int remap_scan_nan_layout(const void *x, int32_t x_dtype, int64_t n_rows,
                          int64_t n_batch, int64_t k_inner,
                          int64_t x_row_stride, int64_t x_batch_stride,
                          int32_t *kinds, void *stream)
{
    return remap::scan_nan_layout(x, x_dtype, n_rows, n_batch, k_inner,
                                  x_row_stride, x_batch_stride, kinds,
                                  static_cast<hipStream_t>(stream));
}

int remap_stream_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    return remap::stream_copy(dst, src, bytes,
                              static_cast<hipStream_t>(stream));
}

int remap_clock_probe(int64_t *ticks_out, int32_t micros, void *stream)
{
    return remap::clock_probe(ticks_out, micros,
                              static_cast<hipStream_t>(stream));
}

}  // extern "C"
