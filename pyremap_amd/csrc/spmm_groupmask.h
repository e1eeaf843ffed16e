// spmm_groupmask.h -- family 10, masked mode of the 8-row groups: the
// normaliser kept per ROW while every source cell of the group is valid or
// missing in ALL of the wave's columns.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// The masked mode (remap_numpy.py:262-266) sums `den = A . [not isnan X]`
// beside `num = A . [X, NaN -> 0]`.  spmm_rowgroup keeps den per lane and
// element: a second accumulator set (64 VGPRs at two K tiles: 202 in all, two
// waves per SIMD -- so it runs ONE tile per wave, twice the waves, twice the
// per-wave overhead) and a third VALU instruction per product.  But a source
// cell is usually valid in every column a wave holds, or missing in every
// one (land, an ice shelf: the whole cell; most groups of a real field see no
// NaN at all).  While that holds, every lane's den is the SAME number: the
// sequential sum of the weights of the row's valid entries.  It is kept in
// one register pair per row (`den_u`), one add per (entry, member) instead of
// one per element, two K tiles per wave as in the frac_b mode.  Validity is
// tested once per entry: two `v_cmp_u_f64` over the lane's four elements, a
// scalar OR, a scalar branch.
//
// The first entry that is valid in some lanes or elements and missing in
// others (a 3-D field cut by bathymetry) sends the GROUP to the general
// form: it starts over, one K tile at a time, with per-lane normalisers --
// the registers of the fast form are free by then.  Nothing is assumed about
// the data; the sums are spmm_rowgroup's, in ascending column order, with the
// same separate multiply and add: same bits.  (A cell missing in every
// column adds `a * 0.0` to num and to den: nothing, for finite `a` -- skipped
// behind a scalar test of the weight's exponent; a NaN or Inf weight sends
// the group to the general form as well.  One exception, under REMAP_FLAG_FMA
// only, which is a tolerance mode anyway: there a sum can be -0.0 -- an
// fma(a, x, +0.0) that underflows -- and adding the skipped `a * 0.0` would
// turn it into +0.0; the sign of such a zero result may differ between the
// two forms.  Without the flag sums start at +0.0 and never are -0.0.)
// ---------------------------------------------------------------------------

// one K tile of one group with per-lane normalisers (spmm_rowgroup's masked
// body at TILES = 1), epilogue included
template <typename XT, bool FMA, int G, int UNR, int VEC>
__device__ __forceinline__ void groupmask_general_tile(
    const KParams &p, const int64_t s, const int64_t woff0, const int64_t e,
    const int32_t *__restrict__ gcol, const double *__restrict__ gw,
    const int32_t *__restrict__ gmask, const int32_t *__restrict__ grid,
    const XT *__restrict__ X, const uint32_t xo, const int64_t yoff_t,
    const bool act_t, const int64_t slot0, const int nmem, const int lane)
{
    typedef typename XVec<XT, VEC>::type xvec_t;
    typedef typename I32Vec<8>::type ivec_t;
    typedef typename I32Vec<G>::type rvec_t;
    double acc[G][1][VEC], den[G][1][VEC];
#pragma unroll
    for (int m = 0; m < G; ++m)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            acc[m][0][v] = 0.0;
            den[m][0][v] = 0.0;
        }
    int64_t woff = woff0;
    for (int64_t base = s; base < e; base += UNR) {
        const int n = (e - base) < UNR ? static_cast<int>(e - base) : UNR;
        const ivec_t cv = *reinterpret_cast<const ivec_t *>(gcol + base);
        const ivec_t mv = *reinterpret_cast<const ivec_t *>(gmask + base);
        const double my_w = gw[woff + lane];
        xvec_t xv[UNR];
#pragma unroll
        for (int uu = 0; uu < UNR; ++uu) {
            if (uu < n) {
                const __amdgpu_buffer_rsrc_t xr = row_rsrc(
                    X + static_cast<int64_t>(cv[uu]) * p.ldx);
                xv[uu] = load_x_buf<XT, VEC>(xr, xo);
            }
        }
        asm volatile("" ::: "memory");
        int idx = 0;
#pragma unroll
        for (int uu = 0; uu < UNR; ++uu) {
            if (uu < n) {
                const int32_t bits = mv[uu];
                double xz[VEC], vf[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double x = elem<xvec_t, VEC>(xv[uu], v);
                    const bool valid = (x == x);
                    xz[v] = valid ? x : 0.0;
                    vf[v] = valid ? 1.0 : 0.0;
                    asm volatile("" : "+v"(xz[v]), "+v"(vf[v]));
                }
#pragma unroll
                for (int m = 0; m < G; ++m) {
                    if (bits & (1 << m)) {
                        const double a = readlane_f64(my_w, idx);
                        ++idx;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            acc[m][0][v] =
                                mul_add<FMA>(a, xz[v], acc[m][0][v]);
                            den[m][0][v] = den_add(a, vf[v], den[m][0][v]);
                        }
                    }
                }
            }
        }
        woff += idx;
    }
    const rvec_t rid = *reinterpret_cast<const rvec_t *>(grid + slot0);
    const bool act1[1] = {act_t};
    const int64_t yoff1[1] = {yoff_t};
#pragma unroll
    for (int m = 0; m < G; ++m) {
        if (m < nmem)
            finish_row<VEC, 1, REMAP_MODE_MASKED>(p, rid[m], 0.0, act1, yoff1,
                                                  acc[m], den[m]);
    }
}

template <typename XT, int TILES, bool FMA, int G, int UNR, int VEC>
__global__ __launch_bounds__(kBlock)
__attribute__((amdgpu_waves_per_eu(3, 8))) void spmm_groupmask(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const int32_t *__restrict__ gcol,
    const double *__restrict__ gw, const int32_t *__restrict__ gmask,
    const int32_t *__restrict__ grid, const double *__restrict__ gfrac,
    const XT *__restrict__ X)
{
    static_assert(UNR <= 8 && UNR * G <= kWave,
                  "a step's weights are one lane-load");
    // union entries in flight in the general form (its second accumulator
    // set must fit the registers the fast form leaves)
    constexpr int kGeneralUnr = TILES == 1 ? 8 : 4;
    typedef typename XVec<XT, VEC>::type xvec_t;
    typedef typename I32Vec<8>::type ivec_t;
    typedef typename I32Vec<G>::type rvec_t;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    REMAP_CLOCK_BEGIN();
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
    uint32_t xo[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        xo[t] = static_cast<uint32_t>(xoff[t] * sizeof(XT));
    // (32 bits: the host checked the grid; a 64-bit count parked in a VGPR
    // pair for want of SGPRs was one of this kernel's two spills)
    const int n_groups_here =
        static_cast<int>((p.row_end - p.row_begin + G - 1) / G);
    const int wpb = static_cast<int>(blockDim.x) >> 6;
    const int block_g0 = static_cast<int>(rb) * (wpb * p.rows_per_wave);

    for (int r = 0; r < p.rows_per_wave; ++r) {
        const int g32 = block_g0 + r * wpb + wave;
        if (g32 >= n_groups_here)
            break;
        const int64_t g = g32;
        const int64_t slot0 = g * G;
        const int nmem = (p.row_end - p.row_begin - slot0) < G
                             ? static_cast<int>(p.row_end - p.row_begin -
                                                slot0)
                             : G;
        const int64_t s = gmeta[2 * g];
        const int64_t woff0 = gmeta[2 * g + 1];
        const int64_t e = gmeta[2 * g + 2];
        bool mixed = false;
        {
            // the fast form: per-row normalisers
            double acc[G][TILES][VEC];
            double den_u[G];
#pragma unroll
            for (int m = 0; m < G; ++m) {
                den_u[m] = 0.0;
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        acc[m][t][v] = 0.0;
            }
            int64_t woff = woff0;
            ivec_t cv = *reinterpret_cast<const ivec_t *>(gcol + s);
            ivec_t mv = *reinterpret_cast<const ivec_t *>(gmask + s);
            for (int64_t base = s; base < e && !mixed; base += UNR) {
                const int n =
                    (e - base) < UNR ? static_cast<int>(e - base) : UNR;
                const double my_w = gw[woff + lane];
                xvec_t xv[UNR][TILES];
#pragma unroll
                for (int uu = 0; uu < UNR; ++uu) {
                    if (uu < n) {
                        const __amdgpu_buffer_rsrc_t xr = row_rsrc(
                            X + static_cast<int64_t>(cv[uu]) * p.ldx);
#pragma unroll
                        for (int t = 0; t < TILES; ++t)
                            xv[uu][t] = load_x_buf<XT, VEC>(xr, xo[t]);
                    }
                }
                // the next step's columns and masks travel meanwhile
                const ivec_t cv_n =
                    *reinterpret_cast<const ivec_t *>(gcol + base + UNR);
                const ivec_t mv_n =
                    *reinterpret_cast<const ivec_t *>(gmask + base + UNR);
                asm volatile("" ::: "memory");
                int idx = 0;
#pragma unroll
                for (int uu = 0; uu < UNR; ++uu) {
                    if (uu < n && !mixed) {
                        const int32_t bits = mv[uu];
                        // lanes holding a NaN among their elements; lanes
                        // whose elements are ALL NaN
                        bool some = false, every = true;
#pragma unroll
                        for (int t = 0; t < TILES; ++t) {
                            if constexpr (VEC == 2) {
                                const double x0 =
                                    elem<xvec_t, VEC>(xv[uu][t], 0);
                                const double x1 =
                                    elem<xvec_t, VEC>(xv[uu][t], 1);
                                some = some || __builtin_isunordered(x0, x1);
                            } else {
                                const double x0 =
                                    elem<xvec_t, VEC>(xv[uu][t], 0);
                                some = some || (x0 != x0);
                            }
                        }
                        const uint64_t some_m = __ballot(some);
                        if (some_m == 0) {
                            // valid in every column: the frac_b mode's
                            // products, and the weight onto the row's den
#pragma unroll
                            for (int m = 0; m < G; ++m) {
                                if (bits & (1 << m)) {
                                    const double a = readlane_f64(my_w, idx);
                                    ++idx;
#pragma unroll
                                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                                        for (int v = 0; v < VEC; ++v)
                                            acc[m][t][v] = mul_add<FMA>(
                                                a,
                                                elem<xvec_t, VEC>(xv[uu][t],
                                                                  v),
                                                acc[m][t][v]);
                                    den_u[m] = den_add(a, 1.0, den_u[m]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int t = 0; t < TILES; ++t)
#pragma unroll
                                for (int v = 0; v < VEC; ++v) {
                                    const double x =
                                        elem<xvec_t, VEC>(xv[uu][t], v);
                                    every = every && (x != x);
                                }
                            if (__ballot(every) != ~0ull) {
                                mixed = true;   // -> the general form
                            } else {
                                // missing in every column: a * 0.0 onto num
                                // and den -- nothing, for a finite weight (a
                                // NaN or Inf one: the general form does the
                                // sums as written)
#pragma unroll
                                for (int m = 0; m < G; ++m) {
                                    if (bits & (1 << m)) {
                                        const int hi =
                                            __builtin_amdgcn_readlane(
                                                __double2hiint(my_w), idx);
                                        if ((hi & 0x7ff00000) == 0x7ff00000)
                                            mixed = true;
                                        ++idx;
                                    }
                                }
                            }
                        }
                    }
                }
                woff += idx;
                cv = cv_n;
                mv = mv_n;
            }
            if (!mixed) {
                const rvec_t rid =
                    *reinterpret_cast<const rvec_t *>(grid + slot0);
#pragma unroll
                for (int m = 0; m < G; ++m) {
                    if (m < nmem) {
                        // the row's normaliser is one number for the whole
                        // wave: the frac_b mode's epilogue (one reciprocal
                        // per row), `den > thr` in place of `frac_b > 0`
                        // (a tile at a time: the results of both tiles
                        // held together cost the four registers this kernel
                        // does not have at three waves per SIMD -- two
                        // spills; the reciprocal is refined once more)
#pragma unroll
                        for (int t = 0; t < TILES; ++t) {
                            const bool act1[1] = {act[t]};
                            const int64_t yoff1[1] = {yoff[t]};
                            double acc1[1][VEC];
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                acc1[0][v] = acc[m][t][v];
                            finish_row_uniform_one_site<VEC, 1>(
                                p, rid[m], den_u[m], den_u[m] > p.thr, act1,
                                yoff1, acc1);
                        }
                    }
                }
            }
        }
        if (mixed) {
            // (one copy of the general body, run once per tile)
#pragma unroll 1
            for (int t = 0; t < TILES; ++t) {
                const bool first = t == 0;
                groupmask_general_tile<XT, FMA, G, kGeneralUnr, VEC>(
                    p, s, woff0, e, gcol, gw, gmask, grid, X,
                    first ? xo[0] : xo[TILES - 1],
                    first ? yoff[0] : yoff[TILES - 1],
                    first ? act[0] : act[TILES - 1], slot0, nmem, lane);
            }
        }
    }
    REMAP_CLOCK_END();
}
