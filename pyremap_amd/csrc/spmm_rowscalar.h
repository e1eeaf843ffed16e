// spmm_rowscalar.h -- family 6: wave per (row, K-chunk), scalar-cache metadata.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
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

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc_sized(
    const void *base, int32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0,
                                             bytes, 0x00020000);
}

// cache policy of the X loads (the builtin's aux operand: bit 0 sc0, bit 1
// nt, bit 4 sc1).  0 in the product; tools/build_diag.py xaux<N> builds the
// A/B variants (profiles/r05_analysis/config5_forms.md)
#ifndef REMAP_X_AUX
#define REMAP_X_AUX 0
#endif

template <typename XT, int VEC>
__device__ __forceinline__ typename XVec<XT, VEC>::type load_x_buf(
    __amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    typedef typename XVec<XT, VEC>::type xvec_t;
    if constexpr (sizeof(xvec_t) == 16) {
        return __builtin_bit_cast(
            xvec_t, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0,
                                                          REMAP_X_AUX));
    } else if constexpr (sizeof(xvec_t) == 8) {
        return __builtin_bit_cast(
            xvec_t, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0,
                                                         REMAP_X_AUX));
    } else {
        return __builtin_bit_cast(
            xvec_t, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0,
                                                         REMAP_X_AUX));
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
                    REMAP_DIAG_COL(p, c);
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
                                den[t][v] = den_add(
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
