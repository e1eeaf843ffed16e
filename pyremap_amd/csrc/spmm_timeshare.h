// spmm_timeshare.h -- family 10, the shared form (spmm_groupshare.h) in the
// masked mode for fields whose mask does not change from batch to batch:
// (Time, nCells, nVertLevels) ocean data cut by bathymetry.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// The masked mode (remap_numpy.py:262-266) sums `den = A . [not isnan X]`
// beside `num = A . [X, NaN -> 0]`, per column: the reference recomputes den
// for every time slice of a mask that depends on (cell, level) only, and so
// does the per-lane form of spmm_rowgroup -- 64 more VGPRs, one K tile per
// wave, config 5 with a bathymetry mask at 26.9 ms where the frac_b mode
// takes 21.  spmm_grouptime.h cut the normalisers to one per lane and row by
// taking four time slices of a level per lane, and stayed at 27.2 ms: beside
// 80 accumulators the registers hold four entries in flight, not eight, and
// the launch is bound by the L1 miss queue, not by the VALU.
//
// The shared form has no entries in flight in registers at all -- the LDS
// ring holds them -- and 12 VGPRs to spare at four waves per SIMD.  Here it
// runs with TIME-MAJOR columns:
//
//   * a workgroup's chunk is (64 levels) x (4 time slices); an entry's piece
//     is 4 x 512 bytes, sent as before by two global_load_lds_dwordx4 (lanes
//     0 - 31: one time slice, 32 - 63: the next): LDS holds [slice][level];
//   * a lane OWNS A LEVEL: it reads its level's four slices (4 ds_read_b64)
//     and keeps four sums and ONE normaliser per row -- while the four
//     slices of every lane are valid together or missing together;
//   * validity once per owned entry (four v_cmp_u_f64, scalar XORs); an
//     entry valid everywhere adds its products with no select.
//
// A wave that meets an entry whose validity differs between the slices of
// some lane keeps sending its pieces and keeping the barriers, and redoes ITS
// group afterwards with per-element normalisers, one slice at a time, from
// global memory (spmm_groupmask.h's general tile).  Nothing is assumed about
// the data: same sums, same order, same bits, with or without
// REMAP_FLAG_BATCH_MASKS.
// ---------------------------------------------------------------------------

template <int OFF>
__device__ __forceinline__ void tshare_read(double &x, uint32_t addr)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2"
                 : "=v"(x)
                 : "v"(addr), "n"(OFF));
}

template <int N>
__device__ __forceinline__ void tshare_wait(double (&x)[4])
{
    asm volatile("s_waitcnt lgkmcnt(%4)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
                 : "n"(N));
}

// x where the lane's bit of `mask` is clear, +0.0 where it is set: two
// v_cndmask_b32 reading the mask from its SGPR pair (written as `mask >> lane
// & 1` hipcc shifts a 64-bit VGPR pair per element; written as `x != x ? 0 :
// x` it compares again).  Inside a branch on the mask: the asm is volatile so
// that the branch is not flattened into selects on every entry.
__device__ __forceinline__ double tshare_zero_where(double x, uint64_t mask)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    asm volatile("v_cndmask_b32_e64 %0, %0, 0, %2\n\t"
                 "v_cndmask_b32_e64 %1, %1, 0, %2"
                 : "+v"(lo), "+v"(hi)
                 : "s"(mask));
    return __hiloint2double(hi, lo);
}

// 1.0 where the lane's bit of `mask` is clear, +0.0 where it is set
__device__ __forceinline__ double tshare_one_where_clear(uint64_t mask)
{
    int hi = 0x3ff00000;
    asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(hi) : "s"(mask));
    return __hiloint2double(hi, 0);
}

template <bool FMA, int AHEAD>
__global__ __launch_bounds__(4 * kWave)
__attribute__((amdgpu_waves_per_eu(4, 8))) void spmm_timeshare(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const int32_t *__restrict__ gcol,
    const double *__restrict__ gw, const int32_t *__restrict__ gmask,
    const int32_t *__restrict__ grid, const int64_t *__restrict__ smeta,
    const int32_t *__restrict__ scol, const int32_t *__restrict__ smask,
    const double *__restrict__ X)
{
    constexpr int G = 8, W = 4, UNR = 8, NBUF = 2, TB = 4;
    constexpr int EPW = UNR / W;
    constexpr int kEntryBytes = TB * 512;    // [slice][64 levels]
    constexpr int kBufBytes = UNR * kEntryBytes;
    constexpr int kWSlot = UNR * G * 8;
    constexpr int kWDma = kWSlot / 256;
    constexpr int kSeg = 2 * kWave;
    static_assert(AHEAD >= 1 && AHEAD * TB <= 15, "LDS reads ahead");
    typedef typename I32Vec<G>::type rvec_t;
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
    // chunk = (block of 64 levels, block of TB time slices), time blocks
    // side by side.  A slice or a level that does not exist is sent from,
    // and read as, the first slice (the row's first value): the same data
    // as an element that does exist, valid or missing with it; never stored.
    const uint32_t n_batch = p.K / p.k_inner;
    const uint32_t n_tb = (n_batch + TB - 1) / TB;
    const uint32_t lb = static_cast<uint32_t>(chunk) / n_tb;
    const uint32_t tb = static_cast<uint32_t>(chunk) - lb * n_tb;
    // the sending side: lane -> (slice 2 t + lane / 32, two levels)
    // (64-bit offsets: the time slices of a (Time, nCells, nVertLevels)
    // field on a 3.7 M-cell mesh are 1.9 GB apart)
    uint64_t xob[2];
    {
        const uint32_t k2 = lb * kWave + 2u * (lane & 31);
        const bool k_on = k2 < p.k_inner;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t b = tb * TB + 2 * t + (lane >> 5);
            const uint32_t bx = b < n_batch ? b : tb * TB;
            xob[t] = k_on ? static_cast<uint64_t>(
                                (static_cast<int64_t>(bx) * p.bsx + k2) * 8)
                          : static_cast<uint64_t>(
                                static_cast<int64_t>(tb * TB) * p.bsx * 8);
        }
    }
    // (the summing side -- lane = level, element e = slice tb * TB + e --
    // needs its offsets only behind the step loop: computed there, 12
    // registers that decide between three and four waves per SIMD)

    const int64_t n_slots = p.row_end - p.row_begin;
    const int64_t n_groups = (n_slots + G - 1) / G;
    const int64_t g = sg * W + wave;
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
    const uint32_t ldx_bytes = static_cast<uint32_t>(p.ldx) * 8u;
    const uint32_t ring_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) char *)ring));
    char *const wring = ring + NBUF * kBufBytes;
    const uint32_t wring_lds = ring_lds + NBUF * kBufBytes;

    double acc[G][TB][1];
    double den_l[G];
#pragma unroll
    for (int m = 0; m < G; ++m) {
        den_l[m] = 0.0;
#pragma unroll
        for (int e = 0; e < TB; ++e)
            acc[m][e][0] = 0.0;
    }
    // lanes that met an entry whose slices disagree: != 0 -> the general form
    uint64_t mixed_bits = 0;

    int seg_w = 0;
    for (int seg0 = 0; seg0 < len; seg0 += kSeg) {
        const int seg_len = (len - seg0) < kSeg ? len - seg0 : kSeg;
        const int seg_steps = (seg_len + UNR - 1) / UNR;
        if (seg0 > 0)
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
        asm volatile("" : : "v"(colv[0]), "v"(colv[1]));
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
                for (int t = 0; t < 2; ++t)
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

            const uint32_t mine = ring_lds + buf * kBufBytes + lane * 8;
            double xr[AHEAD + 1][TB];
            share_static_for(
                std::make_integer_sequence<int, AHEAD>{}, [&](auto d_c) {
                    constexpr int d = decltype(d_c)::value;
                    tshare_read<d * kEntryBytes>(xr[d][0], mine);
                    tshare_read<d * kEntryBytes + 512>(xr[d][1], mine);
                    tshare_read<d * kEntryBytes + 1024>(xr[d][2], mine);
                    tshare_read<d * kEntryBytes + 1536>(xr[d][3], mine);
                });
            share_wait_w<AHEAD * TB>(my_w);
            int idx = 0;
            share_static_for(
                std::make_integer_sequence<int, UNR>{}, [&](auto uu_c) {
                    constexpr int uu = decltype(uu_c)::value;
                    constexpr int slot = uu % (AHEAD + 1);
                    if constexpr (uu + AHEAD < UNR) {
                        constexpr int nx = (uu + AHEAD) % (AHEAD + 1);
                        constexpr int o = (uu + AHEAD) * kEntryBytes;
                        tshare_read<o>(xr[nx][0], mine);
                        tshare_read<o + 512>(xr[nx][1], mine);
                        tshare_read<o + 1024>(xr[nx][2], mine);
                        tshare_read<o + 1536>(xr[nx][3], mine);
                    }
                    const uint32_t word = uu < 4 ? step_lo : step_hi;
                    constexpr int sb = 8 * (uu & 3);
                    if (word & (0xffu << sb)) {
                        constexpr int behind =
                            (uu + AHEAD < UNR ? AHEAD : UNR - 1 - uu) * TB;
                        tshare_wait<behind>(xr[slot]);
                        double x[TB];
#pragma unroll
                        for (int e = 0; e < TB; ++e)
                            x[e] = xr[slot][e];
                        // lanes whose slice e is missing; lanes whose
                        // slices disagree (kept as BITS: tested as a number,
                        // hipcc turns every `^` into s_cmp + s_cselect)
                        uint64_t nan_m[TB];
#pragma unroll
                        for (int e = 0; e < TB; ++e)
                            nan_m[e] = __ballot(x[e] != x[e]);
#pragma unroll
                        for (int e = 1; e < TB; ++e)
                            mixed_bits |= nan_m[e] ^ nan_m[0];
                        if (mixed_bits == 0) {
                            // valid in every lane and slice (the open
                            // ocean): the products as they are; else
                            // missing in some lanes, in all their slices:
                            // those lanes add a * 0.0 to num and to den --
                            // selected by slice 0's mask, which is every
                            // slice's, straight from its SGPR pair
                            double vf = 1.0;
                            if (nan_m[0] != 0) {
#pragma unroll
                                for (int e = 0; e < TB; ++e)
                                    x[e] = tshare_zero_where(x[e], nan_m[0]);
                                vf = tshare_one_where_clear(nan_m[0]);
                            }
#pragma unroll
                            for (int m = 0; m < G; ++m) {
                                if (word & (1u << (sb + m))) {
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
                });
        }
    }

    // the summing side's offsets: lane = level, element e = time slice
    const uint32_t k = lb * kWave + lane;
    const bool lane_on = k < p.k_inner;
    int64_t yoff[TB];
    bool act[TB];
#pragma unroll
    for (int e = 0; e < TB; ++e) {
        const uint32_t b = tb * TB + e;
        act[e] = lane_on && b < n_batch;
        yoff[e] = act[e] ? static_cast<int64_t>(b) * p.bsy + k : 0;
    }
    const bool mixed = mixed_bits != 0;
    if (nmem > 0 && !mixed) {
        const rvec_t rid = *reinterpret_cast<const rvec_t *>(grid + slot0);
#pragma unroll
        for (int m = 0; m < G; ++m) {
            if (m < nmem)
                finish_row_lane_den<TB>(p, rid[m], den_l[m], act, yoff,
                                        acc[m]);
        }
    }
    if (nmem > 0 && mixed) {
        // this wave's group again, with per-element normalisers, one time
        // slice at a time, from global memory (nobody waits for it: the
        // workgroup's last barrier is behind)
        const int64_t s = gmeta[2 * g];
        const int64_t woff0 = gmeta[2 * g + 1];
        const int64_t e_end = gmeta[2 * g + 2];
#pragma unroll 1
        for (int e = 0; e < TB; ++e) {
            // (a slice is one batch: its offset goes into the base pointer,
            // the lane's own offset -- its level -- stays small)
            const uint32_t b_e = tb * TB + e;
            const double *X_e =
                X + static_cast<int64_t>(b_e < n_batch ? b_e : tb * TB) *
                        p.bsx;
            const uint32_t xo_e = lane_on ? k * 8u : 0u;
            const int64_t yoff_e = e == 0   ? yoff[0]
                                   : e == 1 ? yoff[1]
                                   : e == 2 ? yoff[2]
                                            : yoff[3];
            const bool act_e = e == 0   ? act[0]
                               : e == 1 ? act[1]
                               : e == 2 ? act[2]
                                        : act[3];
            groupmask_general_tile<double, FMA, G, 8, 1>(
                p, s, woff0, e_end, gcol, gw, gmask, grid, X_e, xo_e, yoff_e,
                act_e, slot0, nmem, lane);
        }
    }
    REMAP_CLOCK_END();
}
