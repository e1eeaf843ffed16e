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
//  * patch family (opt-in): LDS-staged gather of each destination patch's
//    distinct source rows by LDS-DMA.
//  * rowlane family (K <= 32): one lane per (row, k); lanes of a wave cover
//    64 / K consecutive rows, X accesses are contiguous over k.
//  * Fused epilogue: division by frac_b / by the remapped mask, threshold
//    test, NaN fill and the optional byte mask are applied in registers; the
//    reference's four (n, K) temporaries and its second SpMM never exist.
//  * XCD-aware block map: each XCD (own 4 MiB L2) gets a contiguous range of
//    the chunk-major work list, so the ~nnz/n_a re-touches of a source row by
//    neighbouring destination rows hit that XCD's L2 instead of going back
//    to Infinity Cache / HBM eight times.
#include "remap_common.h"

namespace remap {

char *error_buffer()
{
    static thread_local char buf[kErrorBufferSize] = "";
    return buf;
}

namespace {

struct KParams {
    const int64_t *__restrict__ rowptr;
    const int32_t *__restrict__ col;
    const double *__restrict__ val;
    const void *__restrict__ X;
    double *__restrict__ Y;
    const double *__restrict__ frac_b;
    uint8_t *__restrict__ mask_out;
    const int32_t *__restrict__ row_order;
    int64_t row_begin;
    int64_t row_end;
    int64_t ldx, bsx, ldy, bsy;
    int64_t n_rowblocks;   // row blocks per chunk
    int64_t n_blocks;      // n_rowblocks * n_chunks
    int64_t blocks_per_xcd;
    double thr;
    uint32_t K;
    uint32_t k_inner;
    int32_t rows_per_wave;
    int32_t xcd_map;
    uint32_t x_range;      // bytes addressable from a source row base
    uint32_t y_range;      // bytes addressable from a destination row base
    int32_t debug;         // diagnostics only (tune[6]): 1 = no Y stores,
                           // 2 = gather from the first 1024 source rows
};

template <bool FMA>
__device__ __forceinline__ double mul_add(double a, double x, double acc)
{
    if constexpr (FMA) {
        return __builtin_fma(a, x, acc);
    } else {
        // separate multiply and add (the file is built with
        // -ffp-contract=off): scipy's `y[k] += a * x[k]`
        const double prod = a * x;
        return acc + prod;
    }
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

template <typename XT, int VEC>
struct XVec;
template <>
struct XVec<double, 1> { typedef double type; };
template <>
struct XVec<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <>
struct XVec<float, 1> { typedef float type; };
template <>
struct XVec<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };

template <typename XT, int VEC>
__device__ __forceinline__ typename XVec<XT, VEC>::type load_x(const XT *p)
{
    return *reinterpret_cast<const typename XVec<XT, VEC>::type *>(p);
}

template <typename V, int VEC>
__device__ __forceinline__ double elem(const V &v, int e)
{
    if constexpr (VEC == 1) {
        return static_cast<double>(v);
    } else {
        return static_cast<double>(v[e]);
    }
}

// Y is written once and never read back: non-temporal stores (+3..5 % on
// configs 3 and H against write-back ones, A/B in one run).  There is
// deliberately no run-time switch to plain stores here: with `if (cached)
// plain else nontemporal` on the same address LLVM merges the two stores and
// silently drops the non-temporal hint -- the first builds of this file held
// 414 plain stores and not a single `nt`.
template <int VEC>
__device__ __forceinline__ void store_y(double *p, const double (&y)[VEC])
{
    if constexpr (VEC == 1) {
        __builtin_nontemporal_store(y[0], p);
    } else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 v;
        v[0] = y[0];
        v[1] = y[1];
        __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p));
    }
}

// ---------------------------------------------------------------------------
// pieces shared by the wave-per-row kernels
// ---------------------------------------------------------------------------

// Entries [0, n) of one row chunk are held one per lane in (my_col, my_val).
// Groups of UNROLL entries: every X load of a group is issued before the
// first use, then the group is accumulated strictly in CSR order.
template <typename XT, int VEC, int TILES, int MODE, bool FMA, int UNROLL>
__device__ __forceinline__ void accumulate_entries(
    const XT *__restrict__ X, int64_t ldx, const int64_t (&xoff)[TILES],
    int32_t my_col, double my_val, int n, double (&acc)[TILES][VEC],
    double (&den)[TILES][VEC], int debug = 0)
{
    typedef typename XVec<XT, VEC>::type xvec_t;
    for (int u0 = 0; u0 < n; u0 += UNROLL) {
        xvec_t xv[UNROLL][TILES];
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu) {
            if (u0 + uu < n) {
                int32_t c = __builtin_amdgcn_readlane(my_col, u0 + uu);
                if (debug & 2)
                    c &= 1023;
                const XT *xr = X + static_cast<int64_t>(c) * ldx;
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    xv[uu][t] = load_x<XT, VEC>(xr + xoff[t]);
            }
        }
        asm volatile("" ::: "memory");  // loads stay ahead of their uses
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu) {
            if (u0 + uu < n) {
                const double a = readlane_f64(my_val, u0 + uu);
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double x = elem<xvec_t, VEC>(xv[uu][t], v);
                        if constexpr (MODE == REMAP_MODE_MASKED) {
                            const bool valid = (x == x);
                            const double xz = valid ? x : 0.0;
                            const double mz = valid ? 1.0 : 0.0;
                            acc[t][v] = mul_add<FMA>(a, xz, acc[t][v]);
                            den[t][v] = mul_add<FMA>(a, mz, den[t][v]);
                        } else {
                            acc[t][v] = mul_add<FMA>(a, x, acc[t][v]);
                        }
                    }
            }
        }
    }
}

// Fused epilogue of one row: normalise, mask, store (remap_numpy.py:266-278).
template <int VEC, int TILES, int MODE>
__device__ __forceinline__ void finish_row(
    const KParams &p, int64_t i, double fb, const bool (&act)[TILES],
    const int64_t (&yoff)[TILES], const double (&acc)[TILES][VEC],
    const double (&den)[TILES][VEC])
{
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        if (!act[t])
            continue;
        double y[VEC];
        bool ok[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if constexpr (MODE == REMAP_MODE_RAW) {
                ok[v] = true;
                y[v] = acc[t][v];
            } else if constexpr (MODE == REMAP_MODE_FRACB) {
                // x / 1.0 == x exactly: bilinear maps (frac_b == 1) skip
                // the 11-instruction f64 division; fb is wave-uniform
                ok[v] = fb > 0.0;
                y[v] = !ok[v] ? __builtin_nan("")
                       : (fb == 1.0) ? acc[t][v] : acc[t][v] / fb;
            } else {
                ok[v] = den[t][v] > p.thr;
                y[v] = ok[v] ? acc[t][v] / den[t][v] : __builtin_nan("");
            }
        }
        const int64_t o = i * p.ldy + yoff[t];
        if ((p.debug & 1) && y[0] != 1.2345e300)
            continue;
        store_y<VEC>(p.Y + o, y);
#ifndef REMAP_STAMPS
        if (p.mask_out) {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                p.mask_out[o + v] = ok[v] ? 0 : 1;
        }
#endif
    }
}

// Per-lane element offsets of a wave's K tiles (flat column -> batch, k).
template <int VEC, int TILES>
__device__ __forceinline__ void tile_offsets(
    const KParams &p, int64_t chunk, int lane, int64_t (&xoff)[TILES],
    int64_t (&yoff)[TILES], bool (&act)[TILES])
{
    constexpr int CH = kWave * VEC;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const uint32_t kf = (static_cast<uint32_t>(chunk) * TILES + t) * CH +
                            lane * VEC;
        act[t] = kf < p.K;
        const uint32_t b = act[t] ? kf / p.k_inner : 0u;
        const uint32_t k = act[t] ? kf - b * p.k_inner : 0u;
        // idle lanes (K tail) read offset 0 of the row: harmless, never used
        xoff[t] = static_cast<int64_t>(b) * p.bsx + k;
        yoff[t] = static_cast<int64_t>(b) * p.bsy + k;
    }
}

// physical block -> logical block.  Blocks are dealt round-robin over the 8
// XCDs, so bid % 8 labels the XCD; give each label a contiguous range.
__device__ __forceinline__ int64_t logical_block(const KParams &p)
{
    int64_t L = blockIdx.x;
    if (p.xcd_map) {
        const int64_t xcd = L & (kXcds - 1);
        const int64_t slot = L >> 3;
        L = xcd * p.blocks_per_xcd + slot;
    }
    return L;
}

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(v), src_lane);
    const int hi =
        __builtin_amdgcn_readlane(static_cast<int>(v >> 32), src_lane);
    return (static_cast<int64_t>(hi) << 32) |
           static_cast<int64_t>(static_cast<uint32_t>(lo));
}

// ---------------------------------------------------------------------------
// rowwave: one wave per (row, K-chunk); lanes across K.  Straightforward
// version: each row costs its full dependent chain rowptr -> (col, S) -> X.
// ---------------------------------------------------------------------------
template <typename XT, int VEC, int TILES, int MODE, bool FMA, int UNROLL>
__global__ __launch_bounds__(kBlock) void spmm_rowwave(const KParams p,
                                                       const uint32_t flags)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;  // chunk-major work list
    const int64_t rb = L - chunk * p.n_rowblocks;

    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
    tile_offsets<VEC, TILES>(p, chunk, lane, xoff, yoff, act);

    const XT *__restrict__ X = static_cast<const XT *>(p.X);
    const int64_t block_row0 =
        p.row_begin + rb * (int64_t)(kWavesPerBlock * p.rows_per_wave);

    for (int r = 0; r < p.rows_per_wave; ++r) {
        // the block's waves work on adjacent rows at the same time
        const int64_t slot = block_row0 + (int64_t)r * kWavesPerBlock + wave;
        if (slot >= p.row_end)
            break;
        const int64_t i = p.row_order ? (int64_t)p.row_order[slot] : slot;
        const int64_t s = p.rowptr[i];
        const int64_t e = p.rowptr[i + 1];

        double acc[TILES][VEC];
        double den[TILES][VEC];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[t][v] = 0.0;
                den[t][v] = 0.0;
            }

        for (int64_t base = s; base < e; base += kWave) {
            const int n = (e - base) < kWave ? static_cast<int>(e - base)
                                             : kWave;
            // one coalesced load brings up to 64 (col, S) pairs of the row
            int32_t my_col = 0;
            double my_val = 0.0;
            if (lane < n) {
                my_col = p.col[base + lane];
                my_val = p.val[base + lane];
            }
            accumulate_entries<XT, VEC, TILES, MODE, FMA, UNROLL>(
                X, p.ldx, xoff, my_col, my_val, n, acc, den, p.debug);
        }

        double fb = 0.0;
        if constexpr (MODE == REMAP_MODE_FRACB)
            fb = p.frac_b[i];
        finish_row<VEC, TILES, MODE>(p, i, fb, act, yoff, acc, den);
    }
}

// ---------------------------------------------------------------------------
// patch: LDS-staged gather.  One workgroup owns one PATCH of destination rows
// (a 2-D tile of the destination grid when a row order is installed) x one
// 128-column K-chunk:
//
//   1. gather   every DISTINCT source row the patch references is fetched
//               ONCE, straight into LDS, by LDS-DMA (`global_load_lds_dwordx4`:
//               1 KiB = one row chunk per wave instruction, per-lane source
//               address, no VGPRs, all of a wave's fetches in flight at once);
//   2. barrier  (drains the DMA);
//   3. compute  each wave walks its rows of the patch: (local index, S) pairs
//               come through the SCALAR cache (s_load, 8 entries at a time,
//               no vector-memory instructions), source data from LDS with
//               `ds_read_b128` (lanes across K, sequential sum per lane: the
//               same order and the same bits as the other families), fused
//               epilogue, 16-byte non-temporal stores.
//
// Why: on conservative maps every source row is referenced by nnz/n_a = 3-5
// neighbouring destination rows.  In the register-gather kernels each of
// those references is a separate trip through the CU's vector-memory
// pipeline (texture addresser + L1 miss queue), which is the saturated
// resource; here only distinct rows take that trip and
// the re-touches are LDS reads.  Several workgroups per CU overlap one
// another's gather and compute phases.
//
// Metadata pointers are separate __restrict__ kernel arguments (not members
// of KParams) so hipcc can prove them read-only and use scalar loads.
// ---------------------------------------------------------------------------
constexpr int kPatchBlock = 1024;  // 16 waves
constexpr int kPatchWaves = kPatchBlock / kWave;

// LDS image of one workgroup (row_bytes = 1024 or 512 per staged row chunk):
//   [0, (umax + 1) * row_bytes)  the distinct source-row chunks
//   then                   val  f64[emax]   the patch's weights, slot order
//                          fb   f64[rows]   frac_b of the patch's rows
//                          lidx i32[emax]   their local row indices
//                          rptr i32[rows+1] entry offsets of the patch's rows
//                          rid  i32[rows]   the rows' ids
__host__ __device__ inline uint32_t patch_lds_bytes(int umax, int emax,
                                                    int rows, int row_bytes)
{
    return (static_cast<uint32_t>(umax) + 1u) * row_bytes +
           static_cast<uint32_t>(emax) * 12u +
           static_cast<uint32_t>(rows) * 16u + 32u;
}

// WC = columns per K-chunk: 128 (two doubles per lane, 1 KiB per staged row)
// or 64 (one double per lane, 512 B per staged row: half the LDS per row, so
// twice the patch area fits -- for mappings whose rows reference many
// source rows, e.g. 2nd-order conservative stencils)
template <int MODE, bool FMA, int WC>
__global__ __launch_bounds__(kPatchBlock) void spmm_patch(
    const KParams p, const uint32_t flags,
    const int32_t *__restrict__ prow, const double *__restrict__ pval,
    const int32_t *__restrict__ plidx, const int32_t *__restrict__ pptr,
    const int32_t *__restrict__ ucol, const int32_t *__restrict__ row_order,
    const double *__restrict__ frac_b, const int32_t patch_rows,
    const int32_t umax, const int32_t emax, const int64_t n_patches)
{
    constexpr int VEC = WC / kWave;           // doubles per lane
    constexpr int kRowBytes = WC * 8;         // staged bytes per source row
    constexpr int kRowsPerDma = 1024 / kRowBytes;  // rows per DMA instruction
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef typename XVec<double, VEC>::type xvec_t;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / n_patches;  // chunk-major work list
    const int64_t patch = L - chunk * n_patches;

    // compute-phase columns of this lane
    int64_t xoff[1], yoff[1];
    bool act[1];
    tile_offsets<VEC, 1>(p, chunk, lane, xoff, yoff, act);
    // gather-phase columns: every lane moves 16 B (2 doubles); with 512-byte
    // rows one instruction carries two source rows (lanes 0-31 / 32-63)
    int64_t goff;
    {
        constexpr int kLanesPerRow = kWave / kRowsPerDma;
        const uint32_t kf = static_cast<uint32_t>(chunk) * WC +
                            (lane % kLanesPerRow) * 2;
        const bool in = kf < p.K;
        const uint32_t bb = in ? kf / p.k_inner : 0u;
        const uint32_t kk = in ? kf - bb * p.k_inner : 0u;
        goff = static_cast<int64_t>(bb) * p.bsx + kk;
    }
    const int sub = lane / (kWave / kRowsPerDma);  // which row of the pair

    double *lds_val =
        reinterpret_cast<double *>(lds + (umax + 1) * kRowBytes);
    double *lds_fb = lds_val + emax;
    int32_t *lds_lidx = reinterpret_cast<int32_t *>(lds_fb + patch_rows);
    int32_t *lds_rptr = lds_lidx + emax;
    int32_t *lds_rid = lds_rptr + patch_rows + 1;

    // 1. gather: distinct source rows by LDS-DMA, the patch's entries by
    //    plain loads (they are contiguous: patch-major CSR).  The phase is a
    //    chain of dependent memory trips with every wave of the workgroup
    //    waiting at the barrier behind it, so loads are issued level by
    //    level: everything addressed by the patch id alone first, then what
    //    those values address, LDS writes last (3 trips instead of 5).
    const int u0 = pptr[patch];
    const int U = pptr[patch + 1] - u0;
    const int64_t slot0 = p.row_begin + patch * patch_rows;
    const int64_t local0 = patch * patch_rows;  // index into prow
    int nrows = patch_rows;
    if (slot0 + nrows > p.row_end)
        nrows = static_cast<int>(p.row_end - slot0);
    const int e0 = prow[local0];
    const int n_e = prow[local0 + nrows] - e0;
    // branch-free (clamped) loads: a load inside a divergent branch makes
    // hipcc wait for it on the spot
    const int tc = tid < nrows ? tid : nrows - 1;
    const int32_t *ro = row_order ? row_order + slot0 : prow + local0;
    const int32_t rid_ld = ro[tc];
    const int32_t my_rid =
        row_order ? rid_ld : static_cast<int32_t>(slot0 + tc);
    const int32_t my_rp = prow[local0 + (tid <= nrows ? tid : nrows)];
    constexpr int kPre = 2;  // entry batches held in registers meanwhile
    double ev[kPre];
    int32_t el[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const int t = tid + k * kPatchBlock;
        ev[k] = 0.0;
        el[k] = 0;
        if (t < n_e) {
            ev[k] = pval[e0 + t];
            el[k] = plidx[e0 + t];
        }
    }
    // before the DMA loop: behind it the wait for my_rid would be vmcnt(0)
    double my_fb = 0.0;
    if constexpr (MODE == REMAP_MODE_FRACB)
        my_fb = frac_b[my_rid];
    const double *__restrict__ X = static_cast<const double *>(p.X);
    for (int j = wave * kRowsPerDma; j < U; j += kPatchWaves * kRowsPerDma) {
        // the second row of a pair may not exist: fetch the first again
        // (lands in the spare slot behind the list)
        const int jj = (j + sub < U) ? j + sub : j;
        int32_t c = ucol[u0 + jj];
        if (p.debug & 2)
            c &= 1023;
        const double *g = X + static_cast<int64_t>(c) * p.ldx + goff;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)g,
            (__attribute__((address_space(3))) void *)(lds + j * kRowBytes),
            16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const int t = tid + k * kPatchBlock;
        if (t < n_e) {
            lds_val[t] = ev[k];
            lds_lidx[t] = el[k];
        }
    }
    for (int t = tid + kPre * kPatchBlock; t < n_e; t += kPatchBlock) {
        lds_val[t] = pval[e0 + t];
        lds_lidx[t] = plidx[e0 + t];
    }
    if (tid <= nrows)
        lds_rptr[tid] = my_rp - e0;
    if (tid < nrows) {
        lds_rid[tid] = my_rid;
        if constexpr (MODE == REMAP_MODE_FRACB)
            lds_fb[tid] = my_fb;
    }
    // 2. everything landed, visible to every wave
    __syncthreads();

    // 3. compute the patch's rows from LDS
    const char *mine = lds + lane * (VEC * 8);
    // the next row's header (id, entry range, frac_b) is read while this
    // row is being computed: short rows (4 entries of a bilinear map) are a
    // chain of LDS round trips otherwise
    int32_t nx_rid = 0, nx_s = 0, nx_e = 0;
    double nx_fb = 0.0;
    if (wave < nrows) {
        nx_rid = lds_rid[wave];
        nx_s = lds_rptr[wave];
        nx_e = lds_rptr[wave + 1];
        if constexpr (MODE == REMAP_MODE_FRACB)
            nx_fb = lds_fb[wave];
    }
    for (int r = wave; r < nrows; r += kPatchWaves) {
        const int64_t i = __builtin_amdgcn_readfirstlane(nx_rid);
        const int s = __builtin_amdgcn_readfirstlane(nx_s);
        const int e = __builtin_amdgcn_readfirstlane(nx_e);
        const double fb_row = nx_fb;
        if (r + kPatchWaves < nrows) {
            nx_rid = lds_rid[r + kPatchWaves];
            nx_s = lds_rptr[r + kPatchWaves];
            nx_e = lds_rptr[r + kPatchWaves + 1];
            if constexpr (MODE == REMAP_MODE_FRACB)
                nx_fb = lds_fb[r + kPatchWaves];
        }
        double acc[1][VEC];
        double den[1][VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            acc[0][v] = 0.0;
            den[0][v] = 0.0;
        }
#pragma unroll 4
        for (int jj = s; jj < e; ++jj) {
            // (index, weight) by LDS broadcast (same address in every lane);
            // measured faster than one coalesced read + v_readlane
            const int32_t li = lds_lidx[jj];
            const double a = lds_val[jj];
            const xvec_t xq = *reinterpret_cast<const xvec_t *>(
                mine + li * kRowBytes);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double x = elem<xvec_t, VEC>(xq, v);
                if constexpr (MODE == REMAP_MODE_MASKED) {
                    const bool valid = (x == x);
                    acc[0][v] = mul_add<FMA>(a, valid ? x : 0.0, acc[0][v]);
                    den[0][v] = mul_add<FMA>(a, valid ? 1.0 : 0.0, den[0][v]);
                } else {
                    acc[0][v] = mul_add<FMA>(a, x, acc[0][v]);
                }
            }
        }
        finish_row<VEC, 1, MODE>(p, i, fb_row, act, yoff, acc, den);
    }
}

// ---------------------------------------------------------------------------
// In-kernel stamps (diagnostic build only: -DREMAP_STAMPS, tools/stamps.sh).
// Each stamp reads s_memtime and drains the scalar counter (the recipe of
// cdna_hip_programming.md section 7); the VM flavour first drains vmcnt.  The
// five phase sums of a wave go to a buffer of their own (the launch passes
// it in KParams::mask_out, the byte mask being unused then); no output value
// depends on them.  The product build compiles all of this away.
// ---------------------------------------------------------------------------
#ifdef REMAP_STAMPS
#define REMAP_STAMP_INIT()                                                   \
    unsigned long long st_prev = 0, st_now = 0;                              \
    unsigned long long st_sum[5] = {0, 0, 0, 0, 0};                          \
    unsigned long long st_rows = 0
#define REMAP_STAMP(k)                                                       \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)"                  \
                     : "=s"(st_now)::"memory");                              \
        __builtin_amdgcn_sched_barrier(0);                                   \
        if ((k) != 0)                                                        \
            st_sum[k] += st_now - st_prev;                                   \
        else                                                                 \
            st_rows += 1;                                                    \
        st_prev = st_now;                                                    \
    } while (0)
#define REMAP_STAMP_VM(k)                                                    \
    do {                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
        REMAP_STAMP(k);                                                      \
    } while (0)
#define REMAP_STAMP_FLUSH()                                                  \
    do {                                                                     \
        if (lane == 0 && p.mask_out) {                                       \
            unsigned long long *o =                                          \
                reinterpret_cast<unsigned long long *>(p.mask_out);          \
            for (int k = 1; k < 5; ++k)                                      \
                atomicAdd(o + k, st_sum[k]);                                 \
            atomicAdd(o, st_rows);                                           \
        }                                                                    \
    } while (0)
#else
#define REMAP_STAMP_INIT() do { } while (0)
#define REMAP_STAMP(k) do { } while (0)
#define REMAP_STAMP_VM(k) do { } while (0)
#define REMAP_STAMP_FLUSH() do { } while (0)
#endif

// ---------------------------------------------------------------------------
// rowscalar: the rowwave decomposition with the row metadata taken through
// the SCALAR cache.  rowptr and the row's first 8 (col, S) pairs arrive with
// three wide s_loads (dwordx4 / x8 / x16) straight into SGPRs: no vector-
// memory instruction and no v_readlane is spent on metadata, so the texture
// addresser -- the saturated unit -- only sees the X
// loads and the Y stores.  Needs `csr_pad >= 8` readable entries behind
// col/val (a row's 8-wide fetch may run past its end) and, like the patch
// kernel, separate __restrict__ pointer arguments so hipcc may use s_load.
// ---------------------------------------------------------------------------
// A wave-uniform pointer pinned in SGPRs.  Without this hipcc folds
// "row base + lane offset" into one 64-bit per-lane address; with it the load
// takes the `saddr + 32-bit voffset` form and needs no address VGPR pair.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// X loads through a buffer descriptor built from the (scalar) row base: the
// per-lane part of the address is ONE loop-invariant 32-bit VGPR (voffset),
// so no 64-bit address is formed per load -- fewer VALU instructions, and
// hipcc can no longer recycle a load's destination registers for its address
// (which forced a vmcnt(0) before every load in the masked variant).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0,
                                             0x7fffffff, 0x00020000);
}

template <typename XT, int VEC>
__device__ __forceinline__ typename XVec<XT, VEC>::type load_x_buf(
    __amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    typedef typename XVec<XT, VEC>::type xvec_t;
    if constexpr (sizeof(xvec_t) == 16) {
        return __builtin_bit_cast(
            xvec_t, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
    } else if constexpr (sizeof(xvec_t) == 8) {
        return __builtin_bit_cast(
            xvec_t, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
    } else {
        return __builtin_bit_cast(
            xvec_t, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
    }
}

typedef int32_t i32x8 __attribute__((ext_vector_type(8), aligned(4)));
typedef double f64x8 __attribute__((ext_vector_type(8), aligned(8)));

template <typename XT, int VEC, int TILES, int MODE, bool FMA>
__global__ __launch_bounds__(kBlock) void spmm_rowscalar(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const int32_t *__restrict__ row_order,
    const double *__restrict__ frac_b, const XT *__restrict__ X)
{
    typedef typename XVec<XT, VEC>::type xvec_t;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;
    const int64_t rb = L - chunk * p.n_rowblocks;

    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
    tile_offsets<VEC, TILES>(p, chunk, lane, xoff, yoff, act);
    // 32-bit per-lane element offsets (the host checked that they fit): the
    // loads can then take the scalar row base + 32-bit VGPR offset form and
    // need no 64-bit address registers
    uint32_t xo[TILES];  // BYTE offsets
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        xo[t] = static_cast<uint32_t>(xoff[t] * sizeof(XT));
    const int64_t block_row0 =
        p.row_begin + rb * (int64_t)(kWavesPerBlock * p.rows_per_wave);
    REMAP_STAMP_INIT();

    for (int r = 0; r < p.rows_per_wave; ++r) {
        const int64_t slot = block_row0 + (int64_t)r * kWavesPerBlock + wave;
        if (slot >= p.row_end)
            break;
        const int64_t i = row_order ? (int64_t)row_order[slot] : slot;
        REMAP_STAMP(0);
        const int64_t s = rowptr[i];
        const int64_t e = rowptr[i + 1];
        REMAP_STAMP(1);  // row pointers arrived

        double acc[TILES][VEC];
        double den[TILES][VEC];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[t][v] = 0.0;
                den[t][v] = 0.0;
            }

        for (int64_t base = s; base < e; base += 8) {
            const int n = (e - base) < 8 ? static_cast<int>(e - base) : 8;
            // 8 entries at once through the scalar cache (padded arrays)
            const i32x8 c8 = *reinterpret_cast<const i32x8 *>(col + base);
            const f64x8 a8 = *reinterpret_cast<const f64x8 *>(val + base);
            REMAP_STAMP(2);  // entries arrived
            xvec_t xv[8][TILES];
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
                if (uu < n) {
                    int32_t c = c8[uu];
                    if (p.debug & 2)
                        c &= 1023;
                    const __amdgpu_buffer_rsrc_t xr =
                        row_rsrc(X + static_cast<int64_t>(c) * p.ldx);
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
                        xv[uu][t] = load_x_buf<XT, VEC>(xr, xo[t]);
                }
            }
            // Keep every load of the group issued BEFORE the first use: with
            // X known read-only hipcc otherwise sinks each load next to its
            // use (load, vmcnt(0), compute, load, ...), serialising the row.
            asm volatile("" ::: "memory");
            REMAP_STAMP_VM(3);  // X data arrived
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
                if (uu < n) {
                    const double a = a8[uu];
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const double x = elem<xvec_t, VEC>(xv[uu][t], v);
                            if constexpr (MODE == REMAP_MODE_MASKED) {
                                const bool valid = (x == x);
                                acc[t][v] = mul_add<FMA>(
                                    a, valid ? x : 0.0, acc[t][v]);
                                den[t][v] = mul_add<FMA>(
                                    a, valid ? 1.0 : 0.0, den[t][v]);
                            } else {
                                acc[t][v] = mul_add<FMA>(a, x, acc[t][v]);
                            }
                        }
                }
            }
        }

        double fb = 0.0;
        if constexpr (MODE == REMAP_MODE_FRACB)
            fb = frac_b[i];
        finish_row<VEC, TILES, MODE>(p, i, fb, act, yoff, acc, den);
        REMAP_STAMP(4);  // accumulated, divided, stores issued
    }
    REMAP_STAMP_FLUSH();
}

// ---------------------------------------------------------------------------
// rowgroup: one wave computes 8 destination rows at once (8 consecutive work
// slots: a 2 x 4 tile of a 2-D destination grid) over the sorted UNION of
// their columns.  Each distinct source-row chunk is loaded ONCE per group and
// feeds up to 8 accumulators; neighbouring rows share most of their source
// rows (2-3x fewer loads on wide stencils), and the L1-fill stream -- the
// resource that bounds the entry-rich mappings --
// shrinks by that factor.  Every row still adds its own entries in ascending
// column order: bit-identical to the other families.
//
// Per step of 8 union entries: columns and presence masks through the scalar
// cache (2 x s_load_dwordx8), the 8 x 8 weights with ONE coalesced vector
// load (lane = entry * 8 + member) broadcast by v_readlane with constant lane
// numbers, X via buffer descriptors as in rowscalar.
// ---------------------------------------------------------------------------
constexpr int kGroup = 8;

template <typename XT, int TILES, int MODE, bool FMA>
__global__ __launch_bounds__(kBlock) void spmm_rowgroup(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol,
    const double *__restrict__ gw, const int32_t *__restrict__ gmask,
    const int32_t *__restrict__ row_order, const double *__restrict__ frac_b,
    const XT *__restrict__ X)
{
    constexpr int VEC = 2;
    typedef typename XVec<XT, VEC>::type xvec_t;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;
    const int64_t rb = L - chunk * p.n_rowblocks;

    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
    tile_offsets<VEC, TILES>(p, chunk, lane, xoff, yoff, act);
    uint32_t xo[TILES];  // BYTE offsets (the host checked that they fit)
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        xo[t] = static_cast<uint32_t>(xoff[t] * sizeof(XT));
    const int64_t n_groups_here =
        (p.row_end - p.row_begin + kGroup - 1) / kGroup;
    const int64_t block_g0 = rb * (int64_t)(kWavesPerBlock * p.rows_per_wave);

    for (int r = 0; r < p.rows_per_wave; ++r) {
        const int64_t g = block_g0 + (int64_t)r * kWavesPerBlock + wave;
        if (g >= n_groups_here)
            break;
        const int64_t slot0 = p.row_begin + g * kGroup;
        const int nmem = (p.row_end - slot0) < kGroup
                             ? static_cast<int>(p.row_end - slot0) : kGroup;
        // member m <-> lane m: row id and frac_b of the group's rows
        int32_t my_rid = 0;
        double my_fb = 0.0;
        if (lane < nmem) {
            my_rid = row_order ? row_order[slot0 + lane]
                               : static_cast<int32_t>(slot0 + lane);
            if constexpr (MODE == REMAP_MODE_FRACB)
                my_fb = frac_b[my_rid];
        }
        const int64_t s = gptr[g];
        const int64_t e = gptr[g + 1];

        double acc[kGroup][TILES][VEC];
        double den[kGroup][TILES][VEC];
#pragma unroll
        for (int m = 0; m < kGroup; ++m)
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    acc[m][t][v] = 0.0;
                    den[m][t][v] = 0.0;
                }

        for (int64_t base = s; base < e; base += 8) {
            const int n = (e - base) < 8 ? static_cast<int>(e - base) : 8;
            const i32x8 c8 = *reinterpret_cast<const i32x8 *>(gcol + base);
            const i32x8 m8 = *reinterpret_cast<const i32x8 *>(gmask + base);
            // weights of 8 union entries x 8 members: lane = entry * 8 + m
            const double my_w = gw[base * kGroup + lane];
            xvec_t xv[8][TILES];
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
                if (uu < n) {
                    int32_t c = c8[uu];
                    if (p.debug & 2)
                        c &= 1023;
                    const __amdgpu_buffer_rsrc_t xr =
                        row_rsrc(X + static_cast<int64_t>(c) * p.ldx);
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
                        xv[uu][t] = load_x_buf<XT, VEC>(xr, xo[t]);
                }
            }
            asm volatile("" ::: "memory");  // loads stay ahead of their uses
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
                if (uu < n) {
                    const int32_t bits = m8[uu];
#pragma unroll
                    for (int m = 0; m < kGroup; ++m) {
                        if (bits & (1 << m)) {
                            const double a =
                                readlane_f64(my_w, uu * kGroup + m);
#pragma unroll
                            for (int t = 0; t < TILES; ++t)
#pragma unroll
                                for (int v = 0; v < VEC; ++v) {
                                    const double x =
                                        elem<xvec_t, VEC>(xv[uu][t], v);
                                    if constexpr (MODE == REMAP_MODE_MASKED) {
                                        const bool valid = (x == x);
                                        acc[m][t][v] = mul_add<FMA>(
                                            a, valid ? x : 0.0, acc[m][t][v]);
                                        den[m][t][v] = mul_add<FMA>(
                                            a, valid ? 1.0 : 0.0,
                                            den[m][t][v]);
                                    } else {
                                        acc[m][t][v] =
                                            mul_add<FMA>(a, x, acc[m][t][v]);
                                    }
                                }
                        }
                    }
                }
            }
        }

#pragma unroll
        for (int m = 0; m < kGroup; ++m) {
            if (m < nmem) {
                const int64_t i = __builtin_amdgcn_readlane(my_rid, m);
                double fb = 0.0;
                if constexpr (MODE == REMAP_MODE_FRACB)
                    fb = readlane_f64(my_fb, m);
                finish_row<VEC, TILES, MODE>(p, i, fb, act, yoff, acc[m],
                                             den[m]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// rowlane: one lane per (row, k), for K <= 32
// ---------------------------------------------------------------------------
template <typename XT, int MODE, bool FMA>
__global__ __launch_bounds__(kBlock) void spmm_rowlane(const KParams p,
                                                       const uint32_t flags)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t r = gid / p.K;
    const uint32_t kf = static_cast<uint32_t>(gid - r * p.K);
    if (p.row_begin + r >= p.row_end)
        return;
    const int64_t i = p.row_order ? (int64_t)p.row_order[p.row_begin + r]
                                  : p.row_begin + r;
    const uint32_t b = kf / p.k_inner;
    const uint32_t k = kf - b * p.k_inner;
    const XT *__restrict__ X =
        static_cast<const XT *>(p.X) + (int64_t)b * p.bsx + k;
    const int64_t s = p.rowptr[i];
    const int64_t e = p.rowptr[i + 1];
    double acc = 0.0, den = 0.0;
#pragma unroll 4
    for (int64_t jj = s; jj < e; ++jj) {
        const double a = p.val[jj];
        const double x = static_cast<double>(X[(int64_t)p.col[jj] * p.ldx]);
        if constexpr (MODE == REMAP_MODE_MASKED) {
            const bool valid = (x == x);
            acc = mul_add<FMA>(a, valid ? x : 0.0, acc);
            den = mul_add<FMA>(a, valid ? 1.0 : 0.0, den);
        } else {
            acc = mul_add<FMA>(a, x, acc);
        }
    }
    bool ok = true;
    double y = acc;
    if constexpr (MODE == REMAP_MODE_FRACB) {
        const double fb = p.frac_b[i];
        ok = fb > 0.0;
        y = ok ? acc / fb : __builtin_nan("");
    } else if constexpr (MODE == REMAP_MODE_MASKED) {
        ok = den > p.thr;
        y = ok ? acc / den : __builtin_nan("");
    }
    const int64_t o = i * p.ldy + (int64_t)b * p.bsy + k;
    (void)flags;
    __builtin_nontemporal_store(y, p.Y + o);
    if (p.mask_out)
        p.mask_out[o] = ok ? 0 : 1;
}

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

template <typename XT>
kernel_fn pick_rowlane(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowlane<XT, REMAP_MODE_RAW, true>
                   : spmm_rowlane<XT, REMAP_MODE_RAW, false>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowlane<XT, REMAP_MODE_FRACB, true>
                   : spmm_rowlane<XT, REMAP_MODE_FRACB, false>;
    default:
        return fma ? spmm_rowlane<XT, REMAP_MODE_MASKED, true>
                   : spmm_rowlane<XT, REMAP_MODE_MASKED, false>;
    }
}

typedef void (*patch_fn)(const KParams, const uint32_t, const int32_t *,
                         const double *, const int32_t *, const int32_t *,
                         const int32_t *, const int32_t *, const double *,
                         const int32_t, const int32_t, const int32_t,
                         const int64_t);

template <int WC>
patch_fn pick_patch_wc(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_patch<REMAP_MODE_RAW, true, WC>
                   : spmm_patch<REMAP_MODE_RAW, false, WC>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_patch<REMAP_MODE_FRACB, true, WC>
                   : spmm_patch<REMAP_MODE_FRACB, false, WC>;
    default:
        return fma ? spmm_patch<REMAP_MODE_MASKED, true, WC>
                   : spmm_patch<REMAP_MODE_MASKED, false, WC>;
    }
}

patch_fn pick_patch(int mode, bool fma, int row_bytes)
{
    return row_bytes == 512 ? pick_patch_wc<64>(mode, fma)
                            : pick_patch_wc<128>(mode, fma);
}

// LDS a workgroup may ask for and still leave room for a second one per CU
constexpr uint32_t kPatchLdsMax = 160 * 1024;

bool patch_usable(const remap_apply_args *a, int64_t K64, bool f32,
                  bool can_vec2)
{
    return a->patch_ptr && a->patch_ucol && a->patch_lidx &&
           a->patch_rowptr && a->patch_val && a->patch_rows > 0 &&
           a->patch_rows < kPatchBlock && a->n_patches > 0 &&
           a->patch_umax >= 0 && a->patch_emax >= 0 && !f32 && can_vec2 &&
           K64 >= 2 &&
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
        return pick_rowscalar_mode<XT, 1, 1>(mode, fma);
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

template <typename XT>
struct GroupFn {
    typedef void (*type)(const KParams, const uint32_t, const int64_t *,
                         const int32_t *, const double *, const int32_t *,
                         const int32_t *, const double *, const XT *);
};

template <typename XT, int TILES>
typename GroupFn<XT>::type pick_rowgroup_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowgroup<XT, TILES, REMAP_MODE_RAW, true>
                   : spmm_rowgroup<XT, TILES, REMAP_MODE_RAW, false>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowgroup<XT, TILES, REMAP_MODE_FRACB, true>
                   : spmm_rowgroup<XT, TILES, REMAP_MODE_FRACB, false>;
    default:
        return fma ? spmm_rowgroup<XT, TILES, REMAP_MODE_MASKED, true>
                   : spmm_rowgroup<XT, TILES, REMAP_MODE_MASKED, false>;
    }
}

template <typename XT>
int launch_rowgroup(const remap_apply_args *a, const KParams &p, int tiles,
                    bool fma, int64_t grid, hipStream_t stream)
{
    typename GroupFn<XT>::type fn =
        tiles == 1 ? pick_rowgroup_mode<XT, 1>(a->mode, fma)
                   : pick_rowgroup_mode<XT, 2>(a->mode, fma);
    // tune[7]: KiB of (unused) dynamic LDS per block -- an occupancy throttle
    const uint32_t lds_bytes = a->tune[7] > 0 ? a->tune[7] * 1024u : 0u;
    if (lds_bytes > 64 * 1024)
        REMAP_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(fn),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)), dim3(kBlock),
                       lds_bytes,
                       stream, p, a->flags, a->group_ptr, a->group_col,
                       a->group_w, a->group_mask, a->row_order, a->frac_b,
                       static_cast<const XT *>(a->X));
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
    bool small_offsets;   // byte offsets inside a row fit 32 bits
    bool patch_ok;        // a usable patch plan is attached
    bool group_ok;        // a usable row-group schedule is attached
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
    if (c.K >= (int64_t(1) << 31))
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: K = %lld fields per call exceeds 2^31",
                    (long long)c.K);
    c.fma = (a->flags & REMAP_FLAG_FMA) != 0;
    c.f32 = a->x_dtype == REMAP_DTYPE_F32;
    const size_t xelem = c.f32 ? 4 : 8;
    c.can_vec2 =
        (a->k_inner % 2 == 0) && (a->x_row_stride % 2 == 0) &&
        (a->x_batch_stride % 2 == 0) && (a->y_row_stride % 2 == 0) &&
        (a->y_batch_stride % 2 == 0) && aligned(a->X, 2 * xelem) &&
        aligned(a->Y, 16);
    // per-lane byte offset of the last flat column, from a row base
    c.small_offsets =
        ((a->n_batch - 1) * a->x_batch_stride + a->k_inner) *
            (int64_t)xelem < (int64_t(1) << 31);
    c.patch_ok = patch_usable(a, c.K, c.f32, c.can_vec2);
    c.group_ok = a->group_ptr && a->group_col && a->group_w &&
                 a->group_mask &&
                 a->n_groups == (c.n_rows + kGroup - 1) / kGroup;
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
    p.debug = a->tune[6];
    p.x_range = p.y_range = 0;
    p.row_begin = a->row_begin;
    p.row_end = a->row_end;
    p.ldx = a->x_row_stride;
    p.bsx = a->x_batch_stride;
    p.ldy = a->y_row_stride;
    p.bsy = a->y_batch_stride;
    p.thr = a->threshold;
    p.K = static_cast<uint32_t>(c.K);
    p.k_inner = static_cast<uint32_t>(a->k_inner);
    p.n_rowblocks = p.n_blocks = p.blocks_per_xcd = 0;
    p.rows_per_wave = 0;
    p.xcd_map = 0;
    return p;
}

// REMAP_FLAG_TUNE_HINT: can the preferred family serve this call?
bool hint_usable(const remap_apply_args *a, const Call &c)
{
    if (c.K <= 32)
        return false;  // the lane-per-(row, k) kernel owns small K
    switch (a->tune[0]) {
    case 10:
        return c.group_ok && c.can_vec2 && c.small_offsets;
    case 5:
        return c.patch_ok;
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
    if (c.K <= 32)
        return 2;
    if (c.patch_ok && c.K >= 64)
        return 5;
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

int run_rowgroup(const remap_apply_args *a, const Call &c, KParams p,
                 hipStream_t stream)
{
    if (!c.group_ok || !c.can_vec2 || !c.small_offsets)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the rowgroup kernel needs the "
                    "row-group schedule for [row_begin, row_end), even "
                    "strides and 32-bit offsets");
    // f32 rows are half as long: two K tiles per wave keep the bytes per
    // wave and row at 1 KiB (measured +7 % on config 3 with f32 fields)
    int tiles = a->tune[2] == 0 ? (c.f32 ? 2 : 1) : a->tune[2];
    if (tiles != 2 || c.K <= 128)
        tiles = 1;
    const int gpw = a->tune[3] > 0 ? a->tune[3] : 2;   // groups per wave
    p.rows_per_wave = gpw;
    int64_t grid;
    const int rc = shape_grid(
        p, ceil_div(a->n_groups, (int64_t)kWavesPerBlock * gpw),
        ceil_div(c.K, (int64_t)kWave * 2 * tiles), a->tune[4] != 1, grid);
    if (rc != REMAP_OK)
        return rc;
    return c.f32 ? launch_rowgroup<float>(a, p, tiles, c.fma, grid, stream)
                 : launch_rowgroup<double>(a, p, tiles, c.fma, grid, stream);
}

int run_patch(const remap_apply_args *a, const Call &c, KParams p,
              hipStream_t stream)
{
    if (!c.patch_ok)
        return fail(REMAP_ERR_ARG,
                    "remap_apply_f64: the patch kernel needs a patch plan "
                    "covering [row_begin, row_end), float64 X and even "
                    "strides");
    int64_t grid;
    const int rc = shape_grid(p, a->n_patches,
                              ceil_div(c.K, a->patch_row_bytes / 8),
                              a->tune[4] == 0 || a->tune[4] == 2, grid);
    if (rc != REMAP_OK)
        return rc;
    uint32_t lds_bytes = patch_lds_bytes(a->patch_umax, a->patch_emax,
                                         a->patch_rows, a->patch_row_bytes);
    if (a->tune[7] > 0 && (uint32_t)a->tune[7] * 1024u > lds_bytes)
        lds_bytes = a->tune[7] * 1024u;  // occupancy experiments
    if (lds_bytes < 1024)
        lds_bytes = 1024;
    patch_fn pf = pick_patch(a->mode, c.fma, a->patch_row_bytes);
    if (lds_bytes > 64 * 1024)
        REMAP_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(pf),
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(pf, dim3(static_cast<uint32_t>(grid)),
                       dim3(kPatchBlock), lds_bytes, stream, p, a->flags,
                       a->patch_rowptr, a->patch_val, a->patch_lidx,
                       a->patch_ptr, a->patch_ucol, a->row_order, a->frac_b,
                       a->patch_rows, a->patch_umax, a->patch_emax,
                       a->n_patches);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// tune[7]: KiB of (unused) dynamic LDS per block -- an occupancy throttle for
// experiments: 160 KiB per CU / this = resident blocks per CU
int launch_plain(kernel_fn fn, const remap_apply_args *a, const KParams &p,
                 int64_t grid, hipStream_t stream)
{
    const uint32_t lds_bytes = a->tune[7] > 0 ? a->tune[7] * 1024u : 0u;
    if (lds_bytes > 64 * 1024)
        REMAP_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(fn),
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
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
    return launch_plain(c.f32 ? pick_rowlane<float>(a->mode, c.fma)
                              : pick_rowlane<double>(a->mode, c.fma),
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
    if (vec == 1)
        tiles = 1;
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
        ceil_div(c.K, (int64_t)kWave * vec * tiles),
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

int remap_stream_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    return remap::stream_copy(dst, src, bytes,
                              static_cast<hipStream_t>(stream));
}

}  // extern "C"
