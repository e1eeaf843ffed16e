// spmm_rowgroup.h -- family 10: G destination rows per wave over the union of their columns.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// rowgroup: one wave computes G (8 or 4) destination rows at once -- G
// consecutive work slots: a 2 x 4 (2 x 2) tile of a 2-D destination grid --
// over the sorted UNION of their columns.  Each distinct source-row chunk is
// loaded ONCE per group and feeds up to G accumulators; neighbouring rows
// share most of their source rows (2-3x fewer loads on wide stencils), and
// the L1-fill stream -- the resource that bounds the entry-rich mappings
//  -- shrinks by that factor.  Every row still adds its
// own entries in ascending column order: bit-identical to the other families.
//
// Schedule layout (built once per mapping, remap_apply_args.group_*):
//   group_meta  int64 pairs (first union entry, first weight) per group
//   group_col   the union's source rows, ascending within a group
//   group_mask  bit m set = the group's m-th row owns this union entry
//   group_w     the weights of the PRESENT (union entry, member) pairs only,
//               in (entry, member) order: exactly nnz doubles -- a dense
//               G-wide block per union entry cost 64 B where ~13 B are real
//   group_rid   row id of every work slot, group_frac its frac_b
//
// Per step of UNR union entries: columns and presence masks through the
// scalar cache (s_load_dwordx8), the step's weights with ONE coalesced vector
// load (lane j = j-th present pair of the step) handed to the scalar side by
// v_readlane with a running scalar index, X via buffer descriptors as in
// rowscalar.  Row ids and frac_b of the G rows arrive by two wide s_loads.
// ---------------------------------------------------------------------------
template <int N>
struct I32Vec;
template <>
struct I32Vec<4> {
    typedef int32_t type __attribute__((ext_vector_type(4), aligned(4)));
};
template <>
struct I32Vec<8> {
    typedef int32_t type __attribute__((ext_vector_type(8), aligned(4)));
};
template <>
struct I32Vec<16> {
    typedef int32_t type __attribute__((ext_vector_type(16), aligned(4)));
};
template <int N>
struct F64Vec;

template <>
struct F64Vec<4> {
    typedef double type __attribute__((ext_vector_type(4), aligned(8)));
};
template <>
struct F64Vec<8> {
    typedef double type __attribute__((ext_vector_type(8), aligned(8)));
};
template <>
struct F64Vec<16> {
    typedef double type __attribute__((ext_vector_type(16), aligned(8)));
};

// LOCK (diagnostic build only, tune[5] >= 100; measured and NOT adopted, see
// profiles/r03_analysis/lockstep.md): the waves of a workgroup -- the groups
// of one SUPERGROUP, a 4 x 4 or 8 x 8 tile of the destination grid -- walk
// step-ALIGNED lists (every step covers the same range of source-row ids in
// every wave; the lists carry padding entries with an empty member mask,
// tools/lockstep.py) behind a workgroup barrier per step, so that a source
// row two groups share is requested by both inside the same step: the
// second request is merged with the first in L2 (or hits L1) instead of
// arriving ~20 us later, after the line was evicted.
template <typename XT, int TILES, int MODE, bool FMA, int G, int UNR, int VEC,
          bool LOCK = false, int BLOCK = kBlock>
__global__ __launch_bounds__(BLOCK) void spmm_rowgroup(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const int32_t *__restrict__ gcol,
    const double *__restrict__ gw, const int32_t *__restrict__ gmask,
    const int32_t *__restrict__ grid, const double *__restrict__ gfrac,
    const XT *__restrict__ X)
{
    constexpr bool kPrefetch = (G >= 8 && UNR <= 8);
    typedef typename XVec<XT, VEC>::type xvec_t;
    typedef typename I32Vec<UNR>::type ivec_t;
    typedef typename I32Vec<G>::type rvec_t;
    typedef typename F64Vec<G>::type fvec_t;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    REMAP_CLOCK_BEGIN();
    // work list chunk-major (the K-chunks of a row block far apart: an XCD
    // works on one chunk) or, xcd_map & 2, chunk-minor (the chunks of a row
    // block side by side: its schedule is fetched once per XCD)
    int64_t chunk, rb;
    if (p.xcd_map & 2) {
        const int64_t n_chunks = p.n_blocks / p.n_rowblocks;
        rb = L / n_chunks;
        chunk = L - rb * n_chunks;
    } else {
        chunk = L / p.n_rowblocks;
        rb = L - chunk * p.n_rowblocks;
    }

    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
    tile_offsets<VEC, TILES>(p, chunk, lane, xoff, yoff, act);
    uint32_t xo[TILES];  // BYTE offsets (the host checked that they fit)
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        xo[t] = static_cast<uint32_t>(xoff[t] * sizeof(XT));
    const int64_t n_groups_here = (p.row_end - p.row_begin + G - 1) / G;
    // waves per workgroup: 4, or fewer (entry-rich mappings run 4 % faster
    // with single-wave workgroups: finer-grained dispatch)
    const int wpb = static_cast<int>(blockDim.x) >> 6;
    const int64_t block_g0 = rb * (int64_t)(wpb * p.rows_per_wave);

    for (int r = 0; r < p.rows_per_wave; ++r) {
        const int64_t g = block_g0 + (int64_t)r * wpb + wave;
        if (g >= n_groups_here)
            break;
        const int64_t slot0 = g * G;  // relative to row_begin
        const int nmem = (p.row_end - p.row_begin - slot0) < G
                             ? static_cast<int>(p.row_end - p.row_begin -
                                                slot0)
                             : G;
        const int64_t s = gmeta[2 * g];
        int64_t woff = gmeta[2 * g + 1];
        const int64_t e = gmeta[2 * g + 2];

        double acc[G][TILES][VEC];
        double den[G][TILES][VEC];
#pragma unroll
        for (int m = 0; m < G; ++m)
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    acc[m][t][v] = 0.0;
                    den[m][t][v] = 0.0;
                }

        // one step: UNR union entries, their X loads all in flight, then the
        // present (entry, member) pairs in order
        auto step = [&](const ivec_t &cv, const ivec_t &mv, const int n,
                        ivec_t &cv_next, ivec_t &mv_next,
                        const int64_t next_base) {
            // the step's present weights, one per lane, in (entry, member)
            // order (lanes past the step's count read the next step's)
            constexpr int NW = (UNR * G + kWave - 1) / kWave;
            double my_w[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q)
                my_w[q] = gw[woff + q * kWave + lane];
            xvec_t xv[UNR][TILES];
#pragma unroll
            for (int uu = 0; uu < UNR; ++uu) {
                if (uu < n) {
                    int32_t c = cv[uu];
                    REMAP_DIAG_COL(p, c);
                    // (padding entries of the aligned lists: a descriptor of
                    // zero bytes -- the load returns zeros without a request)
                    const __amdgpu_buffer_rsrc_t xr =
                        LOCK ? row_rsrc_sized(
                                   X + static_cast<int64_t>(c) * p.ldx,
                                   mv[uu] ? 0x7fffffff : 0)
                             : row_rsrc(X + static_cast<int64_t>(c) * p.ldx);
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
                        xv[uu][t] = load_x_buf<XT, VEC>(xr, xo[t]);
                }
            }
            // 8-row groups (3-4 waves per SIMD: every trip shows): the NEXT
            // step's columns and masks travel while this step's X data is
            // awaited (the arrays are padded: always in bounds).  A/B on one
            // box: config 5 22.2 -> 21.8 ms; 4-row groups at 8 waves per
            // SIMD do not gain (config 3 0.360 -> 0.363) and keep the loads
            // at the top of their step.
            if constexpr (kPrefetch) {
                cv_next = *reinterpret_cast<const ivec_t *>(gcol + next_base);
                mv_next = *reinterpret_cast<const ivec_t *>(gmask + next_base);
            }
            asm volatile("" ::: "memory");  // loads stay ahead of their uses
            int idx = 0;  // scalar: next weight of the step
#pragma unroll
            for (int uu = 0; uu < UNR; ++uu) {
                if (uu < n) {
                    const int32_t bits = mv[uu];
                    // masked mode, 8-row groups (entry-rich mappings: an
                    // entry feeds ~4 member rows): the entry's values with
                    // NaN -> 0 and its validity as 1.0 / 0.0 ONCE per entry,
                    // reused by every member row that owns it.  Written
                    // inside the member blocks the compare and three selects
                    // were repeated per member: 8 VALU instructions per
                    // product where 3 do (config 5 masked 31.9 -> 27.9 ms);
                    // the empty asm keeps hipcc from sinking them back.  In
                    // 4-row groups an entry has ~1.1 owners and the 4 extra
                    // VGPRs cost a wave per SIMD: left in the member blocks.
                    constexpr bool kHoist =
                        MODE == REMAP_MODE_MASKED && G >= 8;
                    double xz[TILES][VEC], vf[TILES][VEC];
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const double x = elem<xvec_t, VEC>(xv[uu][t], v);
                            if constexpr (kHoist) {
                                const bool valid = (x == x);
                                xz[t][v] = valid ? x : 0.0;
                                vf[t][v] = valid ? 1.0 : 0.0;
                                asm volatile(""
                                             : "+v"(xz[t][v]), "+v"(vf[t][v]));
                            } else {
                                xz[t][v] = x;
                                vf[t][v] = 0.0;
                            }
                        }
#pragma unroll
                    for (int m = 0; m < G; ++m) {
                        if (bits & (1 << m)) {
                            double a;
                            if constexpr (NW == 1)
                                a = readlane_f64(my_w[0], idx);
                            else
                                a = idx < kWave
                                        ? readlane_f64(my_w[0], idx)
                                        : readlane_f64(my_w[1], idx - kWave);
                            ++idx;
#pragma unroll
                            for (int t = 0; t < TILES; ++t)
#pragma unroll
                                for (int v = 0; v < VEC; ++v) {
                                    if constexpr (kHoist) {
                                        acc[m][t][v] = mul_add<FMA>(
                                            a, xz[t][v], acc[m][t][v]);
                                        den[m][t][v] = den_add(
                                            a, vf[t][v], den[m][t][v]);
                                    } else if constexpr (MODE ==
                                                         REMAP_MODE_MASKED) {
                                        const double x = xz[t][v];
                                        const bool valid = (x == x);
                                        acc[m][t][v] = mul_add<FMA>(
                                            a, valid ? x : 0.0, acc[m][t][v]);
                                        den[m][t][v] = den_add(
                                            a, valid ? 1.0 : 0.0,
                                            den[m][t][v]);
                                    } else {
                                        acc[m][t][v] = mul_add<FMA>(
                                            a, xz[t][v], acc[m][t][v]);
                                    }
                                }
                        }
                    }
                }
            }
            woff += idx;
        };

        if constexpr (kPrefetch) {
            ivec_t cv = *reinterpret_cast<const ivec_t *>(gcol + s);
            ivec_t mv = *reinterpret_cast<const ivec_t *>(gmask + s);
            for (int64_t base = s; base < e; base += UNR) {
                const int n =
                    (e - base) < UNR ? static_cast<int>(e - base) : UNR;
                ivec_t cv_n, mv_n;
                step(cv, mv, n, cv_n, mv_n, base + UNR);
                cv = cv_n;
                mv = mv_n;
            }
        } else {
            for (int64_t base = s; base < e; base += UNR) {
                const int n =
                    (e - base) < UNR ? static_cast<int>(e - base) : UNR;
                const ivec_t cv =
                    *reinterpret_cast<const ivec_t *>(gcol + base);
                const ivec_t mv =
                    *reinterpret_cast<const ivec_t *>(gmask + base);
                if constexpr (LOCK)
                    __builtin_amdgcn_s_barrier();
                ivec_t cv_n, mv_n;
                step(cv, mv, n, cv_n, mv_n, base + UNR);
            }
        }

        // the group's row ids and frac_b, in slot order (padded to whole
        // groups by the host): two wide scalar loads.  (Gathering them, the
        // list bounds and the first 8 entries in one header record per group
        // -- one scalar trip instead of four -- was built and measured:
        // 0.3691 vs 0.3683 ms on config 3, no gain; removed.)
        const rvec_t rid = *reinterpret_cast<const rvec_t *>(grid + slot0);
        fvec_t fbv;
        if constexpr (MODE == REMAP_MODE_FRACB)
            fbv = *reinterpret_cast<const fvec_t *>(gfrac + slot0);
#pragma unroll
        for (int m = 0; m < G; ++m) {
            if (m < nmem) {
                const int64_t i = rid[m];
                double fb = 0.0;
                if constexpr (MODE == REMAP_MODE_FRACB)
                    fb = fbv[m];
                finish_row<VEC, TILES, MODE>(p, i, fb, act, yoff, acc[m],
                                             den[m]);
            }
        }
    }
    REMAP_CLOCK_END();
}
