// spmm_rowgroup.h -- family 10: 8 destination rows per wave over the union of their columns.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
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
                    REMAP_DIAG_COL(p, c);
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
