// spmm_cellshare.h -- family 10, the shared form (spmm_groupshare.h) in the
// masked mode for fields whose cells are missing WHOLE (land, an ice shelf,
// a regional product: every column of the cell, or none):
// REMAP_FLAG_CELL_MASKS on a mapping with the shared lists.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// The masked mode (remap_numpy.py:262-266) sums `den = A . [not isnan X]`
// beside `num = A . [X, NaN -> 0]`.  While every source cell is valid in all
// of a wave's columns or missing in all of them, every lane's den is the same
// number, the sequential sum of the weights of the row's valid entries:
// spmm_groupmask.h keeps it in ONE register pair per row on the 8-row groups
// (config 5 with a quarter of the cells missing: 27.4 -> 22.4 ms).  This is
// the same normaliser on the shared form's decomposition -- a 4-wave
// workgroup per (4 x 8 tile of destination rows) x (256 columns), ONE union
// of source rows per tile through the two-buffer LDS ring, every
// vector-memory instruction of the loop an LDS-DMA:
//
//   * validity once per OWNED entry: two v_cmp_u_f64 over the lane's four
//     elements, a scalar OR; an entry valid everywhere adds the frac_b mode's
//     products and its weight onto the row's den (one add per (entry,
//     member)); an entry missing everywhere adds `a * 0.0` to num and den --
//     nothing, for a finite weight: skipped behind a scalar test of the
//     weight's exponent;
//   * the row's normaliser is wave-uniform: the frac_b mode's epilogue (one
//     reciprocal per row, finish_row_uniform) with `den > thr` in place of
//     `frac_b > 0`.
//
// A wave that meets an entry valid in some lanes or elements and missing in
// others (a field cut by bathymetry under a wrong hint), or a NaN / Inf
// weight on a missing cell, keeps sending its pieces and keeping the barriers
// and redoes ITS group afterwards with per-element normalisers, one K tile at
// a time, from global memory -- flat 64-bit addresses: the batches of a
// (Time, nCells, nVertLevels) field are further apart than a buffer offset
// reaches.  Nothing is assumed about the data: same sums, same order, same
// bits, with or without the flag.
// ---------------------------------------------------------------------------

// one K tile (64 lanes x 2 elements) of one 8-row group with per-element
// normalisers: spmm_groupmask.h's general tile, the source rows through flat
// addresses (a lane's offset from the row's base is 64 bits wide here)
template <bool FMA, int G, int UNR>
__device__ __forceinline__ void cellshare_general_tile(
    const KParams &p, const int64_t s, const int64_t woff0, const int64_t e,
    const int32_t *__restrict__ gcol, const double *__restrict__ gw,
    const int32_t *__restrict__ gmask, const int32_t *__restrict__ grid,
    const double *__restrict__ X, const int64_t xoff_t, const int64_t yoff_t,
    const bool act_t, const int64_t slot0, const int nmem, const int lane)
{
    constexpr int VEC = 2;
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
        share_x2 xv[UNR];
#pragma unroll
        for (int uu = 0; uu < UNR; ++uu) {
            if (uu < n)
                xv[uu] = *reinterpret_cast<const share_x2 *>(
                    X + static_cast<int64_t>(cv[uu]) * p.ldx + xoff_t);
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
                    const double x = xv[uu][v];
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

template <bool FMA, int AHEAD>
__global__ __launch_bounds__(4 * kWave)
__attribute__((amdgpu_waves_per_eu(4, 8))) void spmm_cellshare(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const int32_t *__restrict__ gcol,
    const double *__restrict__ gw, const int32_t *__restrict__ gmask,
    const int32_t *__restrict__ grid, const int64_t *__restrict__ smeta,
    const int32_t *__restrict__ scol, const int32_t *__restrict__ smask,
    const double *__restrict__ X)
{
    constexpr int G = 8, VEC = 2, TILES = 2, W = 4, UNR = 8, NBUF = 2;
    constexpr int EPW = UNR / W;             // entries a wave sends per step
    constexpr int kEntryBytes = TILES * 1024;
    constexpr int kBufBytes = UNR * kEntryBytes;
    constexpr int kWSlot = UNR * G * 8;      // a step's weights at most
    constexpr int kWDma = kWSlot / 256;      // ... 256 bytes per instruction
    constexpr int kSeg = 2 * kWave;          // union entries per segment
    static_assert(AHEAD >= 1 && AHEAD < UNR && AHEAD * TILES <= 15,
                  "LDS reads ahead of the sums");
    typedef typename I32Vec<G>::type rvec_t;
    // NBUF buffers of UNR entries, then NBUF x W slots of weights
    extern __shared__ __attribute__((aligned(16))) char ring[];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    REMAP_CLOCK_BEGIN();
    int64_t chunk, sg;
    if (p.xcd_map & 2) {
        const int64_t n_chunks = p.n_blocks / p.n_rowblocks;
        sg = L / n_chunks;
        chunk = L - sg * n_chunks;
    } else {
        chunk = L / p.n_rowblocks;
        sg = L - chunk * p.n_rowblocks;
    }
    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
    tile_offsets<VEC, TILES>(p, chunk, lane, xoff, yoff, act);

    const int64_t n_slots = p.row_end - p.row_begin;
    const int64_t n_groups = (n_slots + G - 1) / G;
    const int64_t g = sg * W + wave;
    // (a wave past the last group sends its share of the pieces and keeps
    // the barriers; it owns no entry and no row)
    const bool have = g < n_groups;
    const int64_t slot0 = g * G;
    const int nmem = !have ? 0
                     : (n_slots - slot0) < G
                         ? static_cast<int>(n_slots - slot0)
                         : G;
    const int64_t s0 = smeta[2 * sg];
    const int len = static_cast<int>(smeta[2 * sg + 2] - s0);
    const int32_t *__restrict__ lcol = scol + s0;
    const int32_t *__restrict__ lmask = smask + s0;
    const double *__restrict__ lw = gw + gmeta[2 * (have ? g : n_groups) + 1];
    const int sh = wave * G;
    uint64_t xob[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        xob[t] = static_cast<uint64_t>(xoff[t]) * 8u;
    const uint32_t ldx_bytes = static_cast<uint32_t>(p.ldx) * 8u;
    const uint32_t ring_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) char *)ring));
    char *const wring = ring + NBUF * kBufBytes;
    const uint32_t wring_lds = ring_lds + NBUF * kBufBytes;

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
    bool mixed = false;

    int seg_w = 0;   // weights of this wave's stream the earlier segments took
    for (int seg0 = 0; seg0 < len; seg0 += kSeg) {
        const int seg_len = (len - seg0) < kSeg ? len - seg0 : kSeg;
        const int seg_steps = (seg_len + UNR - 1) / UNR;
        if (seg0 > 0)   // the ring of the segment before is read to the end
            share_barrier<0>();
        int32_t colv[2], bitsv[2], bitsh[2], cntv[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            colv[b] = lcol[seg0 + b * kWave + lane];
            const int32_t raw = lmask[seg0 + b * kWave + lane];
            int32_t mine = (raw >> sh) & 0xff;
            mine = seg0 + b * kWave + lane < len ? mine : 0;
            int32_t pc = __builtin_popcount(mine);
            pc += __builtin_amdgcn_update_dpp(0, pc, 0xB1, 0xf, 0xf, true);
            pc += __builtin_amdgcn_update_dpp(0, pc, 0x4E, 0xf, 0xf, true);
            pc += __builtin_amdgcn_update_dpp(0, pc, 0x141, 0xf, 0xf, true);
            // (a step's member bytes packed into its first lane:
            // spmm_groupshare.h)
            share_pack_step<UNR>(mine, bitsv[b], bitsh[b]);
            cntv[b] = pc;
        }
        // (awaited here, in straight-line code: spmm_groupshare.h)
        asm volatile("" : : "v"(colv[0]), "v"(colv[1]));
        // lane j: the weights the steps before step j of the segment took
        int32_t cumv = 0;
        {
            int run = seg_w;
            for (int j = 0; j < seg_steps; ++j) {
                cumv = lane == j ? run : cumv;
                const int e = j * UNR;
                run += __builtin_amdgcn_readlane(
                    e < kWave ? cntv[0] : cntv[1], e & (kWave - 1));
            }
            seg_w = run;
        }

        // this wave's pieces of step st of the segment: its entries of the
        // step and the step's weights
        int32_t col_s = colv[0], bits_lo = bitsv[0], bits_hi = bitsh[0];
        auto send = [&](const int st) {
            const int buf = st % NBUF;
            if (st * UNR == kWave)
                share_switch(col_s, colv[1]);
#pragma unroll
            for (int i = 0; i < EPW; ++i) {
                const int uu = wave * EPW + i;
                int e = st * UNR + uu;
                e = e < seg_len ? e : seg_len - 1;
                int32_t c =
                    __builtin_amdgcn_readlane(col_s, e & (kWave - 1));
                REMAP_DIAG_COL(p, c);
                const char *src =
                    reinterpret_cast<const char *>(X) +
                    static_cast<uint64_t>(static_cast<uint32_t>(c)) *
                        ldx_bytes;
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(
                            src + xob[t]),
                        (__attribute__((address_space(3))) void *)(
                            ring + buf * kBufBytes + uu * kEntryBytes +
                            t * 1024),
                        16, 0, 0);
            }
            const int wo = __builtin_amdgcn_readlane(cumv, st);
            const char *wsrc = reinterpret_cast<const char *>(lw + wo);
#pragma unroll
            for (int q = 0; q < kWDma; ++q)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(
                        wsrc + q * 256 + lane * 4),
                    (__attribute__((address_space(3))) void *)(
                        wring + (buf * W + wave) * kWSlot + q * 256),
                    4, 0, 0);
        };

        if (seg_steps > 0)
            send(0);
        for (int st = 0; st < seg_steps; ++st) {
            const int buf = st % NBUF;
            double my_w;
            share_barrier_w<0>(my_w, wring_lds + (buf * W + wave) * kWSlot +
                                         lane * 8);
            if (st + 1 < seg_steps)
                send(st + 1);
            const int e0 = st * UNR;
            if (e0 == kWave) {
                share_switch(bits_lo, bitsv[1]);
                share_switch(bits_hi, bitsh[1]);
            }
            const uint32_t step_lo = static_cast<uint32_t>(
                __builtin_amdgcn_readlane(bits_lo, e0 & (kWave - 1)));
            const uint32_t step_hi = static_cast<uint32_t>(
                __builtin_amdgcn_readlane(bits_hi, e0 & (kWave - 1)));

            const uint32_t mine = ring_lds + buf * kBufBytes + lane * 16;
            share_x2 xr[AHEAD + 1][TILES];
            share_static_for(
                std::make_integer_sequence<int, AHEAD>{}, [&](auto d_c) {
                    constexpr int d = decltype(d_c)::value;
                    share_read<d * kEntryBytes>(xr[d][0], mine);
                    share_read<d * kEntryBytes + 1024>(xr[d][1], mine);
                });
            share_wait_w<AHEAD * TILES>(my_w);
            int idx = 0;   // scalar: next weight of the step
            share_static_for(
                std::make_integer_sequence<int, UNR>{}, [&](auto uu_c) {
                    constexpr int uu = decltype(uu_c)::value;
                    constexpr int slot = uu % (AHEAD + 1);
                    if constexpr (uu + AHEAD < UNR) {
                        constexpr int nx = (uu + AHEAD) % (AHEAD + 1);
                        share_read<(uu + AHEAD) * kEntryBytes>(xr[nx][0],
                                                               mine);
                        share_read<(uu + AHEAD) * kEntryBytes + 1024>(
                            xr[nx][1], mine);
                    }
                    const uint32_t word = uu < 4 ? step_lo : step_hi;
                    constexpr int sb = 8 * (uu & 3);
                    if (word & (0xffu << sb)) {
                        constexpr int behind =
                            (uu + AHEAD < UNR ? AHEAD : UNR - 1 - uu) *
                            TILES;
                        share_wait<behind, TILES>(xr[slot]);
                        if (!mixed) {
                            // lanes holding a NaN among their four elements
                            const bool some =
                                __builtin_isunordered(xr[slot][0][0],
                                                      xr[slot][0][1]) ||
                                __builtin_isunordered(xr[slot][1][0],
                                                      xr[slot][1][1]);
                            if (__ballot(some) == 0) {
                                // valid in every column: the frac_b mode's
                                // products, the weight onto the row's den
#pragma unroll
                                for (int m = 0; m < G; ++m) {
                                    if (word & (1u << (sb + m))) {
                                        const double a =
                                            readlane_f64(my_w, idx);
                                        ++idx;
#pragma unroll
                                        for (int t = 0; t < TILES; ++t)
#pragma unroll
                                            for (int v = 0; v < VEC; ++v)
                                                acc[m][t][v] = mul_add<FMA>(
                                                    a, xr[slot][t][v],
                                                    acc[m][t][v]);
                                        den_u[m] = den_add(a, 1.0, den_u[m]);
                                    }
                                }
                            } else {
                                bool every = true;
#pragma unroll
                                for (int t = 0; t < TILES; ++t)
#pragma unroll
                                    for (int v = 0; v < VEC; ++v) {
                                        const double x = xr[slot][t][v];
                                        every = every && (x != x);
                                    }
                                if (__ballot(every) != ~0ull) {
                                    mixed = true;   // -> the general form
                                } else {
                                    // missing in every column: a * 0.0 onto
                                    // num and den -- nothing, for a finite
                                    // weight
#pragma unroll
                                    for (int m = 0; m < G; ++m) {
                                        if (word & (1u << (sb + m))) {
                                            const int hi =
                                                __builtin_amdgcn_readlane(
                                                    __double2hiint(my_w),
                                                    idx);
                                            if ((hi & 0x7ff00000) ==
                                                0x7ff00000)
                                                mixed = true;
                                            ++idx;
                                        }
                                    }
                                }
                            }
                        }
                    }
                });
        }
    }

    if (nmem > 0 && !mixed) {
        const rvec_t rid = *reinterpret_cast<const rvec_t *>(grid + slot0);
#pragma unroll
        for (int m = 0; m < G; ++m) {
            if (m < nmem)
                finish_row_uniform<VEC, TILES>(p, rid[m], den_u[m],
                                               den_u[m] > p.thr, act, yoff,
                                               acc[m]);
        }
    }
    if (nmem > 0 && mixed) {
        // this wave's group again, one K tile at a time, from global memory
        // (nobody waits for it: the workgroup's last barrier is behind)
        const int64_t s = gmeta[2 * g];
        const int64_t woff0 = gmeta[2 * g + 1];
        const int64_t e_end = gmeta[2 * g + 2];
#pragma unroll 1
        for (int t = 0; t < TILES; ++t) {
            const bool first = t == 0;
            cellshare_general_tile<FMA, G, 4>(
                p, s, woff0, e_end, gcol, gw, gmask, grid, X,
                first ? xoff[0] : xoff[TILES - 1],
                first ? yoff[0] : yoff[TILES - 1],
                first ? act[0] : act[TILES - 1], slot0, nmem, lane);
        }
    }
    REMAP_CLOCK_END();
}
