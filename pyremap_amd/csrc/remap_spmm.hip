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

template <int VEC>
__device__ __forceinline__ void store_y(double *p, const double (&y)[VEC],
                                        bool cached)
{
    if constexpr (VEC == 1) {
        if (cached)
            *p = y[0];
        else
            __builtin_nontemporal_store(y[0], p);
    } else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 v;
        v[0] = y[0];
        v[1] = y[1];
        if (cached)
            *reinterpret_cast<d2 *>(p) = v;
        else
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
    const double (&den)[TILES][VEC], bool cached)
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
                ok[v] = fb > 0.0;
                y[v] = ok[v] ? acc[t][v] / fb : __builtin_nan("");
            } else {
                ok[v] = den[t][v] > p.thr;
                y[v] = ok[v] ? acc[t][v] / den[t][v] : __builtin_nan("");
            }
        }
        const int64_t o = i * p.ldy + yoff[t];
        if ((p.debug & 1) && y[0] != 1.2345e300)
            continue;
        store_y<VEC>(p.Y + o, y, cached);
        if (p.mask_out) {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                p.mask_out[o + v] = ok[v] ? 0 : 1;
        }
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
    const bool cached = (flags & REMAP_FLAG_CACHED_STORE) != 0;
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
        finish_row<VEC, TILES, MODE>(p, i, fb, act, yoff, acc, den, cached);
    }
}

// ---------------------------------------------------------------------------
// rowpipe: the same work decomposition, software-pipelined.  A wave owns R
// consecutive work slots.
//
//  * It first fetches the row ids, row pointers and frac_b of ALL its rows
//    with one coalesced load each (lane l <-> its l-th row).
//  * (col, S) of a row are prefetched two rows ahead.
//  * Once a row's X data has been consumed into the accumulators, the NEXT
//    row's X loads are issued BEFORE this row's division and stores.  Vector
//    memory operations retire in issue order (s_waitcnt vmcnt counts loads
//    and stores together), so this keeps the stores the youngest outstanding
//    operations: waiting for X data never waits for a store to drain, and
//    the fp64 divisions run under the loads' latency.
//
// TAIL = the K range of this launch has a partial last chunk (lanes may be
// idle); without it the stores are unconditional, which is what lets the
// compiler keep them outstanding across the next wait.
// ---------------------------------------------------------------------------
template <typename XT, int VEC, int TILES, int UNROLL>
__device__ __forceinline__ void issue_group(
    const XT *__restrict__ X, int64_t ldx, const int64_t (&xoff)[TILES],
    int32_t my_col, int n, int debug,
    typename XVec<XT, VEC>::type (&xv)[UNROLL][TILES])
{
#pragma unroll
    for (int uu = 0; uu < UNROLL; ++uu) {
        if (uu < n) {
            int32_t c = __builtin_amdgcn_readlane(my_col, uu);
            if (debug & 2)
                c &= 1023;
            const XT *xr = X + static_cast<int64_t>(c) * ldx;
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                xv[uu][t] = load_x<XT, VEC>(xr + xoff[t]);
        }
    }
}

template <typename XT, int VEC, int TILES, int MODE, bool FMA, int UNROLL>
__device__ __forceinline__ void consume_group(
    const typename XVec<XT, VEC>::type (&xv)[UNROLL][TILES], double my_val,
    int n, double (&acc)[TILES][VEC], double (&den)[TILES][VEC])
{
    typedef typename XVec<XT, VEC>::type xvec_t;
#pragma unroll
    for (int uu = 0; uu < UNROLL; ++uu) {
        if (uu < n) {
            const double a = readlane_f64(my_val, uu);
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

template <typename XT, int VEC, int TILES, int MODE, bool FMA, int UNROLL,
          bool TAIL>
__global__ __launch_bounds__(kBlock) void spmm_rowpipe(const KParams p,
                                                       const uint32_t flags)
{
    typedef typename XVec<XT, VEC>::type xvec_t;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;
    const int64_t rb = L - chunk * p.n_rowblocks;

    const int R = p.rows_per_wave;  // <= 32
    const int64_t slot0 =
        p.row_begin + (rb * kWavesPerBlock + wave) * (int64_t)R;
    if (slot0 >= p.row_end)
        return;
    const int nrows = (p.row_end - slot0) < R
                          ? static_cast<int>(p.row_end - slot0) : R;

    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
    tile_offsets<VEC, TILES>(p, chunk, lane, xoff, yoff, act);
    if constexpr (!TAIL) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            act[t] = true;
    }

    const XT *__restrict__ X = static_cast<const XT *>(p.X);
    const bool cached = (flags & REMAP_FLAG_CACHED_STORE) != 0;

    // metadata of all rows of this wave, one row per lane
    int32_t my_row = 0;
    int64_t my_s = 0, my_e = 0;
    double my_fb = 0.0;
    if (lane < nrows) {
        my_row = p.row_order ? p.row_order[slot0 + lane]
                             : static_cast<int32_t>(slot0 + lane);
        my_s = p.rowptr[my_row];
        my_e = p.rowptr[my_row + 1];
        if constexpr (MODE == REMAP_MODE_FRACB)
            my_fb = p.frac_b[my_row];
    }

    // entries of rows 0 and 1, X data of row 0
    int64_t s = readlane_i64(my_s, 0);
    int64_t e = readlane_i64(my_e, 0);
    int n_cur = (e - s) < kWave ? static_cast<int>(e - s) : kWave;
    int32_t col_cur = 0;
    double val_cur = 0.0;
    if (lane < n_cur) {
        col_cur = p.col[s + lane];
        val_cur = p.val[s + lane];
    }
    int64_t s_next = 0, e_next = 0;
    int n_next = 0;
    int32_t col_next = 0;
    double val_next = 0.0;
    if (nrows > 1) {
        s_next = readlane_i64(my_s, 1);
        e_next = readlane_i64(my_e, 1);
        n_next = (e_next - s_next) < kWave
                     ? static_cast<int>(e_next - s_next) : kWave;
        if (lane < n_next) {
            col_next = p.col[s_next + lane];
            val_next = p.val[s_next + lane];
        }
    }
    xvec_t xv[UNROLL][TILES];
    issue_group<XT, VEC, TILES, UNROLL>(X, p.ldx, xoff, col_cur, n_cur,
                                        p.debug, xv);

    for (int r = 0; r < nrows; ++r) {
        const int64_t i = __builtin_amdgcn_readlane(my_row, r);
        double acc[TILES][VEC];
        double den[TILES][VEC];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[t][v] = 0.0;
                den[t][v] = 0.0;
            }

        // first group of the row: its loads are already in flight
        consume_group<XT, VEC, TILES, MODE, FMA, UNROLL>(xv, val_cur, n_cur,
                                                          acc, den);
        // longer rows: remaining groups, then further 64-entry chunks
        if (n_cur > UNROLL) {
            for (int u0 = UNROLL; u0 < n_cur; u0 += UNROLL) {
                const int32_t c_sh = __shfl(col_cur, lane + u0);
                const double v_sh = __shfl(val_cur, lane + u0);
                accumulate_entries<XT, VEC, TILES, MODE, FMA, UNROLL>(
                    X, p.ldx, xoff, c_sh, v_sh,
                    (n_cur - u0) < UNROLL ? (n_cur - u0) : UNROLL, acc, den,
                    p.debug);
            }
            for (int64_t base = s + kWave; base < e; base += kWave) {
                const int n = (e - base) < kWave ? static_cast<int>(e - base)
                                                 : kWave;
                int32_t my_col = 0;
                double my_val = 0.0;
                if (lane < n) {
                    my_col = p.col[base + lane];
                    my_val = p.val[base + lane];
                }
                accumulate_entries<XT, VEC, TILES, MODE, FMA, UNROLL>(
                    X, p.ldx, xoff, my_col, my_val, n, acc, den, p.debug);
            }
        }

        // the X registers are free again: start the next row's loads and the
        // row-after's entries BEFORE this row's division and stores
        const int32_t col_fin = col_next;
        if (r + 1 < nrows)
            issue_group<XT, VEC, TILES, UNROLL>(X, p.ldx, xoff, col_next,
                                                n_next, p.debug, xv);
        s = s_next;
        e = e_next;
        n_cur = n_next;
        col_cur = col_fin;
        val_cur = val_next;
        n_next = 0;
        if (r + 2 < nrows) {
            s_next = readlane_i64(my_s, r + 2);
            e_next = readlane_i64(my_e, r + 2);
            n_next = (e_next - s_next) < kWave
                         ? static_cast<int>(e_next - s_next) : kWave;
            col_next = 0;
            val_next = 0.0;
            if (lane < n_next) {
                col_next = p.col[s_next + lane];
                val_next = p.val[s_next + lane];
            }
        }

        double fb = 0.0;
        if constexpr (MODE == REMAP_MODE_FRACB)
            fb = readlane_f64(my_fb, r);
        finish_row<VEC, TILES, MODE>(p, i, fb, act, yoff, acc, den, cached);
    }
}

// ---------------------------------------------------------------------------
// rowbuf: the pipelined schedule with a BRANCH-FREE memory instruction
// stream.  Every vector-memory instruction of the row loop is issued
// unconditionally through a buffer descriptor; what must not happen -- an X
// load for an entry the row does not have, a (col, S) prefetch past the
// wave's last row, the byte mask when none was asked for, the K-tail lanes --
// is switched off by the descriptor's range check (num_records = 0, or a
// per-lane offset beyond the range): the hardware drops the access, returns
// zeros, and still counts the instruction.  Because the number of
// outstanding operations is then the same on every path, hipcc can place
// exact `s_waitcnt vmcnt(N)`: the next row's X loads and the row-after's
// entries are in flight under this row's division, and the stores are never
// waited for.  (With `if (entry exists) load` the compiler has to assume the
// shortest path and drains with vmcnt(0) after issuing the next loads.)
//
// LONG = rows may hold more than UNROLL entries (extra groups are handled
// inline, correct but not pipelined).
// ---------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base,
                                                            uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0,
                                             static_cast<int>(bytes),
                                             0x00020000);
}

template <typename XT, int VEC>
struct RawX;
template <>
struct RawX<double, 2> {
    u32x4 v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t o)
    {
        v = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0);
    }
    __device__ __forceinline__ double get(int e) const
    {
        return __hiloint2double(v[2 * e + 1], v[2 * e]);
    }
};
template <>
struct RawX<double, 1> {
    u32x2 v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t o)
    {
        v = __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, 0);
    }
    __device__ __forceinline__ double get(int) const
    {
        return __hiloint2double(v[1], v[0]);
    }
};
template <>
struct RawX<float, 2> {
    u32x2 v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t o)
    {
        v = __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, 0);
    }
    __device__ __forceinline__ double get(int e) const
    {
        return static_cast<double>(__uint_as_float(v[e]));
    }
};
template <>
struct RawX<float, 1> {
    unsigned int v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t o)
    {
        v = __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0);
    }
    __device__ __forceinline__ double get(int) const
    {
        return static_cast<double>(__uint_as_float(v));
    }
};

template <typename XT, int VEC, int TILES, int MODE, bool FMA, int UNROLL,
          bool LONG>
__global__ __launch_bounds__(kBlock) void spmm_rowbuf(const KParams p,
                                                      const uint32_t flags)
{
    constexpr uint32_t kOff = 0xfffffff0u;  // beyond every range: dropped
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;
    const int64_t rb = L - chunk * p.n_rowblocks;

    const int R = p.rows_per_wave;  // <= 32
    const int64_t slot0 =
        p.row_begin + (rb * kWavesPerBlock + wave) * (int64_t)R;
    if (slot0 >= p.row_end)
        return;
    const int nrows = (p.row_end - slot0) < R
                          ? static_cast<int>(p.row_end - slot0) : R;

    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
    tile_offsets<VEC, TILES>(p, chunk, lane, xoff, yoff, act);
    // per-lane BYTE offsets; idle lanes (K tail) point beyond every range
    uint32_t xb[TILES], yb[TILES], mb[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        xb[t] = act[t] ? static_cast<uint32_t>(xoff[t] * sizeof(XT)) : kOff;
        yb[t] = act[t] ? static_cast<uint32_t>(yoff[t] * 8) : kOff;
        mb[t] = act[t] ? static_cast<uint32_t>(yoff[t]) : kOff;
    }

    const char *Xb = static_cast<const char *>(p.X);
    const int64_t ldx_bytes = p.ldx * (int64_t)sizeof(XT);
    const uint32_t mask_range = p.mask_out ? p.y_range / 8 : 0u;
    (void)flags;  // stores are always non-temporal here (no runtime branch)

    // metadata of all rows of this wave, one row per lane
    int32_t my_row = 0;
    int64_t my_s = 0, my_e = 0;
    double my_fb = 0.0;
    if (lane < nrows) {
        my_row = p.row_order ? p.row_order[slot0 + lane]
                             : static_cast<int32_t>(slot0 + lane);
        my_s = p.rowptr[my_row];
        my_e = p.rowptr[my_row + 1];
        if constexpr (MODE == REMAP_MODE_FRACB)
            my_fb = p.frac_b[my_row];
    }

    // (col, S) of a row's first <= 64 entries: lane j <-> entry j; lanes
    // past the row's end (and rows past the wave's last) read zeros
    auto load_entries = [&](int r, int64_t &s_out, int64_t &e_out, int &n_out,
                            int32_t &c_out, double &v_out) {
        const int rr = r < nrows ? r : 0;
        s_out = readlane_i64(my_s, rr);
        e_out = readlane_i64(my_e, rr);
        int n = (e_out - s_out) < kWave ? static_cast<int>(e_out - s_out)
                                        : kWave;
        if (r >= nrows)
            n = 0;
        n_out = n;
        const __amdgpu_buffer_rsrc_t rc =
            make_rsrc(p.col + s_out, static_cast<uint32_t>(n) * 4u);
        const __amdgpu_buffer_rsrc_t rv =
            make_rsrc(p.val + s_out, static_cast<uint32_t>(n) * 8u);
        c_out = static_cast<int32_t>(
            __builtin_amdgcn_raw_buffer_load_b32(rc, lane * 4, 0, 0));
        const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rv, lane * 8, 0, 0);
        v_out = __hiloint2double(w[1], w[0]);
    };

    RawX<XT, VEC> xv[UNROLL][TILES];
    // the first UNROLL entries' X loads, all issued, absent ones dropped
    auto issue_x = [&](int32_t cols, int n) {
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu) {
            int32_t c = __builtin_amdgcn_readlane(cols, uu);
            if (p.debug & 2)
                c &= 1023;
            const __amdgpu_buffer_rsrc_t rx =
                make_rsrc(Xb + (int64_t)c * ldx_bytes,
                          uu < n ? p.x_range : 0u);
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                xv[uu][t].load(rx, xb[t]);
        }
    };

    // division + stores of one row; `live` = false issues the same stores
    // through null descriptors (dropped by the range check)
    auto finish = [&](int64_t i, double fb, const double (&acc)[TILES][VEC],
                      const double (&den)[TILES][VEC], bool live) {
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(
            p.Y + i * p.ldy, (live && !(p.debug & 1)) ? p.y_range : 0u);
        const __amdgpu_buffer_rsrc_t rm =
            make_rsrc(p.mask_out + i * p.ldy, live ? mask_range : 0u);
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            double y[VEC];
            bool ok[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if constexpr (MODE == REMAP_MODE_RAW) {
                    ok[v] = true;
                    y[v] = acc[t][v];
                } else if constexpr (MODE == REMAP_MODE_FRACB) {
                    ok[v] = fb > 0.0;
                    y[v] = ok[v] ? acc[t][v] / fb : __builtin_nan("");
                } else {
                    ok[v] = den[t][v] > p.thr;
                    y[v] = ok[v] ? acc[t][v] / den[t][v] : __builtin_nan("");
                }
            }
            if constexpr (VEC == 2) {
                u32x4 w;
                w[0] = __double2loint(y[0]);
                w[1] = __double2hiint(y[0]);
                w[2] = __double2loint(y[1]);
                w[3] = __double2hiint(y[1]);
                __builtin_amdgcn_raw_buffer_store_b128(w, ry, yb[t], 0, 2);
                const unsigned short m =
                    (ok[0] ? 0 : 1) | ((ok[1] ? 0 : 1) << 8);
                __builtin_amdgcn_raw_buffer_store_b16(m, rm, mb[t], 0, 0);
            } else {
                u32x2 w;
                w[0] = __double2loint(y[0]);
                w[1] = __double2hiint(y[0]);
                __builtin_amdgcn_raw_buffer_store_b64(w, ry, yb[t], 0, 2);
                const unsigned char m = ok[0] ? 0 : 1;
                __builtin_amdgcn_raw_buffer_store_b8(m, rm, mb[t], 0, 0);
            }
        }
    };

    int64_t s, e, s_next, e_next;
    int n_cur, n_next;
    int32_t col_cur, col_next;
    double val_cur, val_next;
    // Prologue shaped like the tail of a loop iteration -- X loads, entry
    // prefetch, stores (null here) -- so the outstanding-operation count at
    // the loop head is the same from both predecessors.
    load_entries(0, s, e, n_cur, col_cur, val_cur);
    issue_x(col_cur, n_cur);
    load_entries(1, s_next, e_next, n_next, col_next, val_next);
    {
        double zero[TILES][VEC];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                zero[t][v] = 0.0;
        finish(0, 1.0, zero, zero, false);
    }

    for (int r = 0; r < nrows; ++r) {
        const int64_t i = __builtin_amdgcn_readlane(my_row, r);
        double acc[TILES][VEC];
        double den[TILES][VEC];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[t][v] = 0.0;
                den[t][v] = 0.0;
            }

        // consume the first group (VALU only inside the uniform branches)
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu) {
            if (uu < n_cur) {
                const double a = readlane_f64(val_cur, uu);
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double x = xv[uu][t].get(v);
                        if constexpr (MODE == REMAP_MODE_MASKED) {
                            const bool valid = (x == x);
                            acc[t][v] = mul_add<FMA>(a, valid ? x : 0.0,
                                                     acc[t][v]);
                            den[t][v] = mul_add<FMA>(a, valid ? 1.0 : 0.0,
                                                     den[t][v]);
                        } else {
                            acc[t][v] = mul_add<FMA>(a, x, acc[t][v]);
                        }
                    }
            }
        }
        if constexpr (LONG) {
            if (e - s > UNROLL) {
                const XT *__restrict__ X = static_cast<const XT *>(p.X);
                for (int u0 = UNROLL; u0 < n_cur; u0 += UNROLL) {
                    const int32_t c_sh = __shfl(col_cur, lane + u0);
                    const double v_sh = __shfl(val_cur, lane + u0);
                    accumulate_entries<XT, VEC, TILES, MODE, FMA, UNROLL>(
                        X, p.ldx, xoff, c_sh, v_sh,
                        (n_cur - u0) < UNROLL ? (n_cur - u0) : UNROLL, acc,
                        den, p.debug);
                }
                for (int64_t base = s + kWave; base < e; base += kWave) {
                    const int n = (e - base) < kWave
                                      ? static_cast<int>(e - base) : kWave;
                    int32_t my_col = 0;
                    double my_val = 0.0;
                    if (lane < n) {
                        my_col = p.col[base + lane];
                        my_val = p.val[base + lane];
                    }
                    accumulate_entries<XT, VEC, TILES, MODE, FMA, UNROLL>(
                        X, p.ldx, xoff, my_col, my_val, n, acc, den, p.debug);
                }
            }
        }

        // next row's X loads and the row-after's entries go out BEFORE this
        // row's division and stores
        issue_x(col_next, n_next);
        s = s_next;
        e = e_next;
        n_cur = n_next;
        col_cur = col_next;
        val_cur = val_next;
        load_entries(r + 2, s_next, e_next, n_next, col_next, val_next);

        // fused epilogue (remap_numpy.py:266-278)
        double fb = 0.0;
        if constexpr (MODE == REMAP_MODE_FRACB)
            fb = readlane_f64(my_fb, r);
        finish(i, fb, acc, den, true);
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
    if (flags & REMAP_FLAG_CACHED_STORE)
        p.Y[o] = y;
    else
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

template <typename XT, int VEC, int TILES, int UNROLL, bool TAIL>
kernel_fn pick_rowpipe_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowpipe<XT, VEC, TILES, REMAP_MODE_RAW, true, UNROLL, TAIL>
                   : spmm_rowpipe<XT, VEC, TILES, REMAP_MODE_RAW, false, UNROLL, TAIL>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowpipe<XT, VEC, TILES, REMAP_MODE_FRACB, true, UNROLL, TAIL>
                   : spmm_rowpipe<XT, VEC, TILES, REMAP_MODE_FRACB, false, UNROLL, TAIL>;
    default:
        return fma ? spmm_rowpipe<XT, VEC, TILES, REMAP_MODE_MASKED, true, UNROLL, TAIL>
                   : spmm_rowpipe<XT, VEC, TILES, REMAP_MODE_MASKED, false, UNROLL, TAIL>;
    }
}

template <typename XT, int VEC, int TILES, int UNROLL>
kernel_fn pick_rowpipe(int mode, bool fma, bool tail)
{
    return tail ? pick_rowpipe_mode<XT, VEC, TILES, UNROLL, true>(mode, fma)
                : pick_rowpipe_mode<XT, VEC, TILES, UNROLL, false>(mode, fma);
}

// tune[5] = entries in flight per group: 0/8 -> 8, 4 -> 4
template <typename XT>
kernel_fn pick_rowpipe_shape(int vec, int tiles, int unroll, int mode,
                             bool fma, bool tail)
{
    if (vec == 1)
        return pick_rowpipe<XT, 1, 1, 8>(mode, fma, tail);
    switch (tiles) {
    case 1:
        return pick_rowpipe<XT, 2, 1, 8>(mode, fma, tail);
    case 2:
        return unroll == 4 ? pick_rowpipe<XT, 2, 2, 4>(mode, fma, tail)
                           : pick_rowpipe<XT, 2, 2, 8>(mode, fma, tail);
    default:
        return unroll == 8 ? pick_rowpipe<XT, 2, 4, 8>(mode, fma, tail)
                           : pick_rowpipe<XT, 2, 4, 4>(mode, fma, tail);
    }
}

template <typename XT, int VEC, int TILES, int UNROLL, bool LONG>
kernel_fn pick_rowbuf_mode(int mode, bool fma)
{
    switch (mode) {
    case REMAP_MODE_RAW:
        return fma ? spmm_rowbuf<XT, VEC, TILES, REMAP_MODE_RAW, true, UNROLL, LONG>
                   : spmm_rowbuf<XT, VEC, TILES, REMAP_MODE_RAW, false, UNROLL, LONG>;
    case REMAP_MODE_FRACB:
        return fma ? spmm_rowbuf<XT, VEC, TILES, REMAP_MODE_FRACB, true, UNROLL, LONG>
                   : spmm_rowbuf<XT, VEC, TILES, REMAP_MODE_FRACB, false, UNROLL, LONG>;
    default:
        return fma ? spmm_rowbuf<XT, VEC, TILES, REMAP_MODE_MASKED, true, UNROLL, LONG>
                   : spmm_rowbuf<XT, VEC, TILES, REMAP_MODE_MASKED, false, UNROLL, LONG>;
    }
}

template <typename XT, int VEC, int TILES, int UNROLL>
kernel_fn pick_rowbuf(int mode, bool fma, bool longrows)
{
    return longrows ? pick_rowbuf_mode<XT, VEC, TILES, UNROLL, true>(mode, fma)
                    : pick_rowbuf_mode<XT, VEC, TILES, UNROLL, false>(mode, fma);
}

template <typename XT>
kernel_fn pick_rowbuf_shape(int vec, int tiles, int unroll, int mode,
                            bool fma, bool longrows)
{
    if (vec == 1)
        return pick_rowbuf<XT, 1, 1, 8>(mode, fma, longrows);
    switch (tiles) {
    case 1:
        return pick_rowbuf<XT, 2, 1, 8>(mode, fma, longrows);
    case 2:
        return unroll == 4 ? pick_rowbuf<XT, 2, 2, 4>(mode, fma, longrows)
                           : pick_rowbuf<XT, 2, 2, 8>(mode, fma, longrows);
    default:
        return pick_rowbuf<XT, 2, 4, 4>(mode, fma, longrows);
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

bool aligned(const void *p, size_t a)
{
    return (reinterpret_cast<uintptr_t>(p) % a) == 0;
}

}  // namespace

int apply(const remap_apply_args *a, hipStream_t stream)
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
    const int64_t K64 = a->n_batch * a->k_inner;
    const int64_t n_rows = a->row_end - a->row_begin;
    if (n_rows == 0 || K64 == 0)
        return REMAP_OK;  // empty output: nothing to launch
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
    if (K64 >= (int64_t(1) << 31))
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: K = %lld fields per call exceeds 2^31",
                    (long long)K64);

    const bool fma = (a->flags & REMAP_FLAG_FMA) != 0;
    const bool f32 = a->x_dtype == REMAP_DTYPE_F32;
    const size_t xelem = f32 ? 4 : 8;

    KParams p;
    p.rowptr = A.rowptr;
    p.col = A.col;
    p.val = A.val;
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
    p.K = static_cast<uint32_t>(K64);
    p.k_inner = static_cast<uint32_t>(a->k_inner);

    int family = a->tune[0];
    if (family == 0)
        family = (K64 <= 32) ? 2 : 1;

    kernel_fn fn = nullptr;
    int64_t grid = 0;
    if (family == 2) {
        fn = f32 ? pick_rowlane<float>(a->mode, fma)
                 : pick_rowlane<double>(a->mode, fma);
        const int64_t threads = n_rows * K64;
        grid = (threads + kBlock - 1) / kBlock;
        p.n_rowblocks = p.n_blocks = p.blocks_per_xcd = 0;
        p.rows_per_wave = 0;
        p.xcd_map = 0;
    } else if (family == 1 || family == 3 || family == 4) {
        // two elements per lane need even strides and aligned bases
        const bool can_vec2 =
            (a->k_inner % 2 == 0) && (p.ldx % 2 == 0) && (p.bsx % 2 == 0) &&
            (p.ldy % 2 == 0) && (p.bsy % 2 == 0) &&
            aligned(a->X, 2 * xelem) && aligned(a->Y, 16);
        int vec = a->tune[1];
        if (vec == 0)
            vec = (can_vec2 && K64 > 64) ? 2 : 1;
        if (vec == 2 && !can_vec2)
            return fail(REMAP_ERR_ARG,
                        "remap_apply_f64: 2 elements per lane need even "
                        "strides and 16-byte aligned X/Y");
        if (vec != 1 && vec != 2)
            return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[1] = %d", vec);
        int tiles = a->tune[2];
        if (tiles == 0)
            tiles = K64 >= 256 ? 2 : 1;  // measured best on config 3
        if (vec == 1)
            tiles = 1;
        if (tiles != 1 && tiles != 2 && tiles != 4)
            return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[2] = %d",
                        tiles);
        int rpw = a->tune[3];
        if (rpw == 0)
            rpw = family >= 3 ? 8 : 4;
        if (rpw < 1 || rpw > (family >= 3 ? 32 : 1024))
            return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[3] = %d", rpw);
        if (family >= 3 && A.n_rows >= (int64_t(1) << 31))
            return fail(REMAP_ERR_UNSUPPORTED,
                        "remap_apply_f64: more than 2^31 rows");
        int map = a->tune[4];
        if (map == 0)
            map = 2;
        const int64_t chunk_cols = (int64_t)kWave * vec * tiles;
        const int64_t n_chunks = (K64 + chunk_cols - 1) / chunk_cols;
        const int64_t rows_per_block = (int64_t)kWavesPerBlock * rpw;
        p.n_rowblocks = (n_rows + rows_per_block - 1) / rows_per_block;
        p.n_blocks = p.n_rowblocks * n_chunks;
        p.rows_per_wave = rpw;
        p.xcd_map = (map == 2) ? 1 : 0;
        p.blocks_per_xcd = (p.n_blocks + kXcds - 1) / kXcds;
        grid = p.xcd_map ? p.blocks_per_xcd * kXcds : p.n_blocks;
        const bool tail = (K64 % chunk_cols) != 0;
        if (family == 4) {
            // buffer addressing: 32-bit byte offsets from a row base
            const int64_t xr = ((a->n_batch - 1) * p.bsx + a->k_inner) *
                               (int64_t)xelem;
            const int64_t yr = ((a->n_batch - 1) * p.bsy + a->k_inner) * 8;
            if (xr >= (int64_t(1) << 31) || yr >= (int64_t(1) << 31))
                return fail(REMAP_ERR_UNSUPPORTED,
                            "remap_apply_f64: batch stride beyond the 2 GiB "
                            "buffer range of the rowbuf kernels");
            p.x_range = static_cast<uint32_t>(xr);
            p.y_range = static_cast<uint32_t>(yr);
            const int unroll = (tiles == 4 || a->tune[5] == 4) ? 4 : 8;
            const bool longrows = A.max_row_nnz <= 0 ||
                                  A.max_row_nnz > (vec == 1 || tiles != 4
                                                   ? (a->tune[5] == 4 && tiles == 2 ? 4 : 8) : 4);
            (void)unroll;
            fn = f32 ? pick_rowbuf_shape<float>(vec, tiles, a->tune[5],
                                                a->mode, fma, longrows)
                     : pick_rowbuf_shape<double>(vec, tiles, a->tune[5],
                                                 a->mode, fma, longrows);
        } else if (family == 3)
            fn = f32 ? pick_rowpipe_shape<float>(vec, tiles, a->tune[5],
                                                 a->mode, fma, tail)
                     : pick_rowpipe_shape<double>(vec, tiles, a->tune[5],
                                                  a->mode, fma, tail);
        else
            fn = f32 ? pick_rowwave_shape<float>(vec, tiles, a->mode, fma)
                     : pick_rowwave_shape<double>(vec, tiles, a->mode, fma);
    } else {
        return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[0] = %d", family);
    }
    if (grid <= 0 || grid > 0x7fffffffLL)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: grid of %lld blocks; split the rows",
                    (long long)grid);

    // tune[7]: KiB of (unused) dynamic LDS per block -- an occupancy throttle
    // for experiments: 160 KiB per CU / this = resident blocks per CU
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
