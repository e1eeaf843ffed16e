// spmm_grouproll.h -- family 10, rolling form: the row-group kernel with the
// source-row loads of the NEXT step issued as the registers of this step fall
// free.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// spmm_rowgroup (spmm_rowgroup.h) works a group's union list in steps of UNR
// entries: issue the step's X loads, wait, sum, issue the next step's.  Two
// things leave a wave's loads and its sums apart in time there:
//   * the loads of a step stand behind `uu < n` tests (the last step of a
//     list is short), so hipcc's wait-count pass must assume the FEWEST loads
//     in flight at every join: the first product of a full step waits behind
//     `s_waitcnt vmcnt(0)` -- for all sixteen loads, not for its own two;
//   * the next step's loads are issued after the last sum of this one: while
//     a wave sums it has nothing in flight, while it waits it sums nothing.
//     Entry-rich mappings run 3 waves per SIMD (144 VGPRs): nothing else
//     covers for it.
// Here every step is FULL as far as the instruction stream goes -- entries
// past the end of the list get a buffer descriptor of zero bytes (the load
// returns zeros without a request) and an empty member mask -- so no load is
// conditional and every wait is exact; and as soon as entry u of step s is
// summed, its registers take entry u of step s + 1: UNR entries stay in
// flight through the whole list.  The last step runs as a copy of the body
// without refills (a conditional refill would bring the joins back).
// The sums are those of spmm_rowgroup, entry by entry, member by member, in
// ascending column order: same bits.
//
// MEASURED (round 5, profiles/r05_analysis/config5_forms.md), and NOT chosen
// by any schedule: config 5 24.2-25.2 ms against 21.9-22.9 (+10 %), config 3
// +2-5 %, headline -1 %, config 5 masked with one K tile -3 %.  The waits are
// exact and the fabric reads fall (42 against 58 GB), but a CU's L1 already
// has its miss queue full (72 requests in flight): loads queued behind a
// wave's own earlier loads lengthen the queue, they do not fill it.  Kept
// behind tune[5] = 26 / 28 (6 / 8 union entries in flight).
// ---------------------------------------------------------------------------
template <typename XT, int TILES, int MODE, bool FMA, int G, int UNR, int VEC>
__global__ __launch_bounds__(kBlock) void spmm_grouproll(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const int32_t *__restrict__ gcol,
    const double *__restrict__ gw, const int32_t *__restrict__ gmask,
    const int32_t *__restrict__ grid, const double *__restrict__ gfrac,
    const XT *__restrict__ X)
{
    static_assert(UNR <= 8 && UNR * G <= 2 * kWave,
                  "a step's weights are one or two lane-loads");
    constexpr int NW = (UNR * G + kWave - 1) / kWave;
    typedef typename XVec<XT, VEC>::type xvec_t;
    typedef typename I32Vec<8>::type ivec_t;
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
        int64_t base = gmeta[2 * g];
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

        // entry u of a step: descriptor (zero bytes past the list's end: no
        // request, zeros back) and the loads of its TILES pieces
        auto fetch = [&](xvec_t (&dst)[TILES], int32_t c, bool on) {
            REMAP_DIAG_COL(p, c);
            const __amdgpu_buffer_rsrc_t xr = row_rsrc_sized(
                X + static_cast<int64_t>(c) * p.ldx, on ? 0x7fffffff : 0);
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                dst[t] = load_x_buf<XT, VEC>(xr, xo[t]);
        };

        // the products of one union entry, member by member (spmm_rowgroup's
        // inner block; `idx` walks the step's lane-held weights)
        auto consume = [&](const int32_t bits, const xvec_t (&x_in)[TILES],
                           const double (&my_w)[NW], int &idx) {
            constexpr bool kHoist = MODE == REMAP_MODE_MASKED && G >= 8;
            double xz[TILES][VEC], vf[TILES][VEC];
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double x = elem<xvec_t, VEC>(x_in[t], v);
                    if constexpr (kHoist) {
                        const bool valid = (x == x);
                        xz[t][v] = valid ? x : 0.0;
                        vf[t][v] = valid ? 1.0 : 0.0;
                        asm volatile("" : "+v"(xz[t][v]), "+v"(vf[t][v]));
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
                                acc[m][t][v] = mul_add<FMA>(a, xz[t][v],
                                                            acc[m][t][v]);
                                den[m][t][v] =
                                    den_add(a, vf[t][v], den[m][t][v]);
                            } else if constexpr (MODE == REMAP_MODE_MASKED) {
                                const double x = xz[t][v];
                                const bool valid = (x == x);
                                acc[m][t][v] = mul_add<FMA>(
                                    a, valid ? x : 0.0, acc[m][t][v]);
                                den[m][t][v] = den_add(
                                    a, valid ? 1.0 : 0.0, den[m][t][v]);
                            } else {
                                acc[m][t][v] = mul_add<FMA>(a, xz[t][v],
                                                            acc[m][t][v]);
                            }
                        }
                }
            }
        };

        // first step: columns, masks, weights, X
        ivec_t cv = *reinterpret_cast<const ivec_t *>(gcol + base);
        ivec_t mv = *reinterpret_cast<const ivec_t *>(gmask + base);
        // (the weights FIRST, here as in the loop: the wait counts at the
        // loop's head are the tighter of its two entries' -- with the weights
        // issued among the X loads the first product of EVERY step waited for
        // all but three of the loads in flight)
        double w[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q)
            w[q] = gw[woff + q * kWave + lane];
        asm volatile("" ::: "memory");
        xvec_t xv[UNR][TILES];
        {
            const int n = (e - base) < UNR ? static_cast<int>(e - base) : UNR;
#pragma unroll
            for (int uu = 0; uu < UNR; ++uu) {
                const bool on = uu < n;
                if (!on)
                    mv[uu] = 0;
                fetch(xv[uu], cv[uu], on);
            }
        }
        asm volatile("" ::: "memory");

        // full steps with a successor: sum entry u, refill its registers
        while (base + UNR < e) {
            const int64_t nbase = base + UNR;
            const ivec_t cvn = *reinterpret_cast<const ivec_t *>(gcol + nbase);
            ivec_t mvn = *reinterpret_cast<const ivec_t *>(gmask + nbase);
            const int nn =
                (e - nbase) < UNR ? static_cast<int>(e - nbase) : UNR;
            int cnt = 0;  // present pairs of this step
#pragma unroll
            for (int uu = 0; uu < UNR; ++uu)
                cnt += __builtin_popcount(static_cast<uint32_t>(mv[uu]));
            double wn[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q)
                wn[q] = gw[woff + cnt + q * kWave + lane];
            asm volatile("" ::: "memory");
            int idx = 0;
#pragma unroll
            for (int uu = 0; uu < UNR; ++uu) {
                consume(mv[uu], xv[uu], w, idx);
                const bool on = uu < nn;
                if (!on)
                    mvn[uu] = 0;
                fetch(xv[uu], cvn[uu], on);
                asm volatile("" ::: "memory");
            }
            woff += cnt;
#pragma unroll
            for (int q = 0; q < NW; ++q)
                w[q] = wn[q];
            mv = mvn;
            base = nbase;
        }
        // last step: nothing to fetch
        {
            int idx = 0;
#pragma unroll
            for (int uu = 0; uu < UNR; ++uu)
                consume(mv[uu], xv[uu], w, idx);
        }

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
