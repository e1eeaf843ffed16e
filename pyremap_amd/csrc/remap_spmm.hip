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
// rowwave: one wave per (row, K-chunk); lanes across K
// ---------------------------------------------------------------------------
template <typename XT, int VEC, int TILES, int MODE, bool FMA, int UNROLL>
__global__ __launch_bounds__(kBlock) void spmm_rowwave(const KParams p,
                                                       const uint32_t flags)
{
    constexpr int CH = kWave * VEC;  // flat columns per tile
    typedef typename XVec<XT, VEC>::type xvec_t;

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // physical block -> logical block.  Blocks are dealt round-robin over the
    // 8 XCDs, so bid % 8 labels the XCD; give each label a contiguous range.
    int64_t L = blockIdx.x;
    if (p.xcd_map) {
        const int64_t xcd = L & (kXcds - 1);
        const int64_t slot = L >> 3;
        L = xcd * p.blocks_per_xcd + slot;
    }
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;  // chunk-major work list
    const int64_t rb = L - chunk * p.n_rowblocks;

    // per-lane element offsets of this wave's K tiles
    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
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

    const XT *__restrict__ X = static_cast<const XT *>(p.X);
    const bool cached = (flags & REMAP_FLAG_CACHED_STORE) != 0;
    const int64_t block_row0 =
        p.row_begin + rb * (int64_t)(kWavesPerBlock * p.rows_per_wave);

    for (int r = 0; r < p.rows_per_wave; ++r) {
        // the block's waves work on adjacent rows at the same time
        const int64_t i = block_row0 + (int64_t)r * kWavesPerBlock + wave;
        if (i >= p.row_end)
            break;
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
            for (int u0 = 0; u0 < n; u0 += UNROLL) {
                xvec_t xv[UNROLL][TILES];
                // issue every load of this group before the first use
#pragma unroll
                for (int uu = 0; uu < UNROLL; ++uu) {
                    if (u0 + uu < n) {
                        const int32_t c =
                            __builtin_amdgcn_readlane(my_col, u0 + uu);
                        const XT *xr = X + static_cast<int64_t>(c) * p.ldx;
#pragma unroll
                        for (int t = 0; t < TILES; ++t)
                            xv[uu][t] = load_x<XT, VEC>(xr + xoff[t]);
                    }
                }
                // accumulate strictly in CSR order
#pragma unroll
                for (int uu = 0; uu < UNROLL; ++uu) {
                    if (u0 + uu < n) {
                        const double a = readlane_f64(my_val, u0 + uu);
#pragma unroll
                        for (int t = 0; t < TILES; ++t)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                const double x =
                                    elem<xvec_t, VEC>(xv[uu][t], v);
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

        // fused epilogue: normalise, mask, store
        double fb = 0.0;
        if constexpr (MODE == REMAP_MODE_FRACB)
            fb = p.frac_b[i];
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
            store_y<VEC>(p.Y + o, y, cached);
            if (p.mask_out) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    p.mask_out[o + v] = ok[v] ? 0 : 1;
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
    const int64_t i = p.row_begin + r;
    if (i >= p.row_end)
        return;
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
    } else if (family == 1) {
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
            tiles = 1;
        if (vec == 1)
            tiles = 1;
        if (tiles != 1 && tiles != 2 && tiles != 4)
            return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[2] = %d",
                        tiles);
        int rpw = a->tune[3];
        if (rpw == 0)
            rpw = 4;
        if (rpw < 1 || rpw > 1024)
            return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[3] = %d", rpw);
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
        fn = f32 ? pick_rowwave_shape<float>(vec, tiles, a->mode, fma)
                 : pick_rowwave_shape<double>(vec, tiles, a->mode, fma);
    } else {
        return fail(REMAP_ERR_ARG, "remap_apply_f64: tune[0] = %d", family);
    }
    if (grid <= 0 || grid > 0x7fffffffLL)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_apply_f64: grid of %lld blocks; split the rows",
                    (long long)grid);

    hipLaunchKernelGGL(fn, dim3(static_cast<uint32_t>(grid)), dim3(kBlock), 0,
                       stream, p, a->flags);
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
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16;
         i += stride)
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
    size_t grid = (n16 + kBlock - 1) / kBlock;
    if (grid > 256 * 8)
        grid = 256 * 8;  // 8 blocks per CU, grid-stride the rest
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
