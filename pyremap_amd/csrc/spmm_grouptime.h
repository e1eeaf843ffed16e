// spmm_grouptime.h -- family 10, masked mode of the 8-row groups for fields
// whose mask does not change from batch to batch: (Time, nCells, nVertLevels)
// ocean data cut by bathymetry.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// The masked mode (remap_numpy.py:262-266) sums `den = A . [not isnan X]`
// beside `num = A . [X, NaN -> 0]`, per column.  A 3-D MPAS-Ocean variable is
// missing below the sea floor: validity depends on (cell, level) and NOT on
// time -- the reference recomputes `den` for every time slice all the same
// (`matrix.dot(in_mask)` over all of time x level, :265), and so did the
// per-lane form of spmm_rowgroup: one normaliser per lane and element, 64
// more VGPRs, a third VALU instruction per product, one K tile per wave
// (config 5 with a bathymetry mask: 26.7 ms where the frac_b mode takes 21).
//
// Here a wave's columns are TIME-MAJOR: lane = level, the lane's TB = 4
// elements = four consecutive time slices of that level (four 8-byte loads
// per entry, each a contiguous run of levels).  While the four elements of a
// lane are valid together or missing together -- a mask that does not depend
// on time -- the lane needs ONE normaliser per row: 16 VGPRs instead of 64,
// one FMA per (entry, member) instead of four, and the den of a (row, level)
// is summed once per four time slices.  Validity is tested once per entry
// (four v_cmp_u_f64, scalar XORs of the lane masks); an entry valid in every
// lane and slice -- the open ocean -- takes the frac_b mode's products with
// no select at all.
//
// The first entry whose validity differs between the time slices of some lane
// sends the GROUP to the general form (spmm_groupmask.h's: per-element
// normalisers, one time slice at a time) -- nothing is assumed about the
// data, the sums are spmm_rowgroup's, in ascending column order with the same
// separate multiply and add: the same bits whatever the field holds and
// whether or not the caller passes REMAP_FLAG_BATCH_MASKS.
// ---------------------------------------------------------------------------
constexpr int kTimeBlock = 4;   // time slices per lane

template <typename XT, bool FMA, int G, int UNR, int LOADS>
__global__ __launch_bounds__(kBlock)
__attribute__((amdgpu_waves_per_eu(3, 8))) void spmm_grouptime(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const int32_t *__restrict__ gcol,
    const double *__restrict__ gw, const int32_t *__restrict__ gmask,
    const int32_t *__restrict__ grid, const double *__restrict__ gfrac,
    const XT *__restrict__ X)
{
    static_assert(UNR == 8 && UNR * G <= kWave && UNR % LOADS == 0,
                  "a step's weights are one lane-load");
    constexpr int TB = kTimeBlock;
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
    // chunk = (block of 64 levels, block of TB time slices), time blocks
    // side by side: lane = level, element e = time slice tb * TB + e.  An
    // element that does not exist (the last time block, levels past the run)
    // reads where element 0 reads -- the same value, valid or missing with
    // it -- and is never stored.
    const uint32_t n_batch = p.K / p.k_inner;
    const uint32_t n_tb = (n_batch + TB - 1) / TB;
    const uint32_t lb = static_cast<uint32_t>(chunk) / n_tb;
    const uint32_t tb = static_cast<uint32_t>(chunk) - lb * n_tb;
    const uint32_t k = lb * kWave + lane;
    const bool lane_on = k < p.k_inner;
    int64_t yoff[TB];
    uint32_t xo[TB];
    bool act[TB];
#pragma unroll
    for (int e = 0; e < TB; ++e) {
        const uint32_t b = tb * TB + e;
        act[e] = lane_on && b < n_batch;
        const uint32_t bx = b < n_batch ? b : tb * TB;
        xo[e] = lane_on ? static_cast<uint32_t>(
                              (static_cast<int64_t>(bx) * p.bsx + k) *
                              static_cast<int64_t>(sizeof(XT)))
                        : 0u;
        yoff[e] = act[e] ? static_cast<int64_t>(b) * p.bsy + k : 0;
    }
    const int64_t n_groups_here = (p.row_end - p.row_begin + G - 1) / G;
    const int wpb = static_cast<int>(blockDim.x) >> 6;
    const int64_t block_g0 = rb * (int64_t)(wpb * p.rows_per_wave);

    for (int r = 0; r < p.rows_per_wave; ++r) {
        const int64_t g = block_g0 + (int64_t)r * wpb + wave;
        if (g >= n_groups_here)
            break;
        const int64_t slot0 = g * G;
        const int nmem = (p.row_end - p.row_begin - slot0) < G
                             ? static_cast<int>(p.row_end - p.row_begin -
                                                slot0)
                             : G;
        const int64_t s = gmeta[2 * g];
        const int64_t woff0 = gmeta[2 * g + 1];
        const int64_t e_end = gmeta[2 * g + 2];
        bool mixed = false;
        {
            // the fast form: one normaliser per lane (level) and row
            double acc[G][TB][1];
            double den_l[G];
#pragma unroll
            for (int m = 0; m < G; ++m) {
                den_l[m] = 0.0;
#pragma unroll
                for (int e = 0; e < TB; ++e)
                    acc[m][e][0] = 0.0;
            }
            int64_t woff = woff0;
            ivec_t cv = *reinterpret_cast<const ivec_t *>(gcol + s);
            ivec_t mv = *reinterpret_cast<const ivec_t *>(gmask + s);
            for (int64_t base = s; base < e_end && !mixed; base += UNR) {
                const int n = (e_end - base) < UNR
                                  ? static_cast<int>(e_end - base)
                                  : UNR;
                const double my_w = gw[woff + lane];
                // the next step's columns and masks travel meanwhile
                const ivec_t cv_n =
                    *reinterpret_cast<const ivec_t *>(gcol + base + UNR);
                const ivec_t mv_n =
                    *reinterpret_cast<const ivec_t *>(gmask + base + UNR);
                int idx = 0;
                // LOADS entries (x TB slices: 2 KiB of a 64-level field) in
                // flight at a time -- what the per-lane form keeps in flight,
                // and what the registers beside 80 accumulators hold
#pragma unroll
                for (int h = 0; h < UNR; h += LOADS) {
                XT xv[LOADS][TB];
#pragma unroll
                for (int uu = 0; uu < LOADS; ++uu) {
                    if (h + uu < n) {
                        const __amdgpu_buffer_rsrc_t xr = row_rsrc(
                            X + static_cast<int64_t>(cv[h + uu]) * p.ldx);
#pragma unroll
                        for (int e = 0; e < TB; ++e)
                            xv[uu][e] = load_x_buf<XT, 1>(xr, xo[e]);
                    }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int uq = 0; uq < LOADS; ++uq) {
                    const int uu = h + uq;
                    if (uu < n && !mixed) {
                        const int32_t bits = mv[uu];
                        double x[TB];
#pragma unroll
                        for (int e = 0; e < TB; ++e)
                            x[e] = static_cast<double>(xv[uq][e]);
                        // lanes whose slice e is missing
                        uint64_t nan_m[TB];
#pragma unroll
                        for (int e = 0; e < TB; ++e)
                            nan_m[e] = __ballot(x[e] != x[e]);
                        uint64_t differ = 0, any = nan_m[0];
#pragma unroll
                        for (int e = 1; e < TB; ++e) {
                            differ |= nan_m[e] ^ nan_m[0];
                            any |= nan_m[e];
                        }
                        if (differ != 0) {
                            mixed = true;   // -> the general form
                        } else {
                            // valid in every lane and slice (the open
                            // ocean): the frac_b mode's products as they
                            // are, the weight onto the dens; else missing in
                            // some lanes, in all their slices: those lanes
                            // add a * 0.0 to num and to den
                            double vf = 1.0;
                            if (any != 0) {
                                const bool valid = x[0] == x[0];
                                vf = valid ? 1.0 : 0.0;
#pragma unroll
                                for (int e = 0; e < TB; ++e)
                                    x[e] = valid ? x[e] : 0.0;
                            }
#pragma unroll
                            for (int e = 0; e < TB; ++e)
                                asm volatile("" : "+v"(x[e]));
                            asm volatile("" : "+v"(vf));
#pragma unroll
                            for (int m = 0; m < G; ++m) {
                                if (bits & (1 << m)) {
                                    const double a = readlane_f64(my_w, idx);
                                    ++idx;
#pragma unroll
                                    for (int e = 0; e < TB; ++e)
                                        acc[m][e][0] = mul_add<FMA>(
                                            a, x[e], acc[m][e][0]);
                                    den_l[m] = den_add(a, vf, den_l[m]);
                                }
                            }
                        }
                    }
                }
                // (the next half's loads stay behind this half's sums)
                asm volatile("" ::: "memory");
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
                        double den[TB][1];
#pragma unroll
                        for (int e = 0; e < TB; ++e)
                            den[e][0] = den_l[m];
                        finish_row<1, TB, REMAP_MODE_MASKED>(
                            p, rid[m], 0.0, act, yoff, acc[m], den);
                    }
                }
            }
        }
        if (mixed) {
            // (one copy of the general body, run once per time slice)
#pragma unroll 1
            for (int e = 0; e < TB; ++e) {
                const uint32_t xo_e = e == 0   ? xo[0]
                                      : e == 1 ? xo[1]
                                      : e == 2 ? xo[2]
                                               : xo[3];
                const int64_t yoff_e = e == 0   ? yoff[0]
                                       : e == 1 ? yoff[1]
                                       : e == 2 ? yoff[2]
                                                : yoff[3];
                const bool act_e = e == 0   ? act[0]
                                   : e == 1 ? act[1]
                                   : e == 2 ? act[2]
                                            : act[3];
                groupmask_general_tile<XT, FMA, G, 8, 1>(
                    p, s, woff0, e_end, gcol, gw, gmask, grid, X, xo_e,
                    yoff_e, act_e, slot0, nmem, lane);
            }
        }
    }
    REMAP_CLOCK_END();
}
