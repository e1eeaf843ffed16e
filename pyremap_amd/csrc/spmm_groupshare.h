// spmm_groupshare.h -- family 10, the SHARED form: W waves, one union, LDS.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// The row-group kernel (spmm_rowgroup.h) loads every distinct source row of a
// wave's 8 destination rows once per wave.  On entry-rich mappings (2nd-order
// conservative: config 5) neighbouring 8-row groups still share most of their
// source rows, every one of them pulls its own copy from L2, and the launch
// is bound by the L1 miss queue of the CU (profiles/r05_analysis/
// config5_forms.md: 218 GB of L1 fills for 30 GB of X).  Larger groups cut
// the fills and lose the occupancy that keeps that queue full: a wave has no
// room for more than 8 rows x 256 columns of accumulators at 3 waves per SIMD.
//
// Here W (2 or 4) waves -- a workgroup -- own W consecutive 8-row groups, a
// SUPERGROUP: a 4 x 4 or 4 x 8 tile of the destination grid (the 8-row
// groups are walked inside super_tile = 4 or 8 blocks, so consecutive groups
// stack up to such a tile).  The supergroup has ONE sorted union of source
// rows (share_col, share_mask: bit 8 w + m = member m of wave w owns the
// entry; remap_share_build).  Per step of UNR union entries:
//
//   * every wave sends its share of the step's entries, 1 KiB per
//     instruction, straight from global memory into an LDS ring by LDS-DMA
//     (global_load_lds_dwordx4) -- each distinct source row enters the CU
//     ONCE per supergroup: 0.16 union entries per entry on config 5 where the
//     8-row groups have 0.33;
//   * one s_barrier; the DMA of step s + 1 flies while step s is summed;
//   * every wave reads the step's entries from LDS (ds_read_b128, a few
//     entries ahead of the sums) and adds the ones its own 8 rows own --
//     exactly the inner loop of spmm_rowgroup: member bits from the mask, the
//     wave's weights (group_w of the 8-row schedule: its own contiguous
//     stream, one coalesced load per step) handed over by v_readlane with a
//     running scalar index.  A row adds its own entries in ascending column
//     order: the same bits as every other family.
//
// The wave keeps no X values in flight in registers (the ring does): 8 rows x
// 256 columns of accumulators + two or three entries on their way from LDS.
// Columns and masks do not pass through the scalar cache here: a scalar load
// in flight turns every LDS wait into lgkmcnt(0); they are fetched 64 entries
// at a time, one per lane, and handed out by v_readlane -- the masks already
// cut down to this wave's 8 member bits, with the number of weights each
// step takes summed over its 8 lanes (three DPP adds per 64 entries): a
// step's scalar side is nine v_readlane (the first build extracted bits and
// counts entry by entry on the scalar unit: 100 SALU instructions per step,
// 5.7e9 per launch against the row-group kernel's 3.5e9, and an on-chip floor
// of 17.7 ms where that kernel has 14.6 -- profiles/r06_analysis).
// ---------------------------------------------------------------------------
template <>
struct I32Vec<2> {
    typedef int32_t type __attribute__((ext_vector_type(2), aligned(4)));
};

// compile-time loop: the body sees its index as a constant (the offsets of
// the ds_read_b128 below are instruction immediates)
template <int... I, typename F>
__device__ __forceinline__ void share_static_for(
    std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}

// every DMA and every load of this wave has landed, every LDS read of the
// last step is done; then the workgroup's barrier
__device__ __forceinline__ void share_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The ring is read with explicit ds_read_b128 / s_waitcnt lgkmcnt(N): left to
// hipcc, every read into a register whose last value was never used (an
// entry this wave's rows do not own) is preceded by `s_waitcnt lgkmcnt(0)`
// -- the reads ahead are drained at every entry that is skipped, and more
// than half of them are.  LDS reads return in order and nothing else in the
// loop counts on lgkmcnt, so N = the reads issued behind the one awaited (a
// scalar load the compiler may add only makes the wait longer, never too
// short).  The "+v" operands tie the uses of the values behind the wait.
typedef double share_x2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ void share_read(share_x2 &x, uint32_t addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2"
                 : "=v"(x)
                 : "v"(addr), "n"(OFF));
}

template <int N, int TILES>
__device__ __forceinline__ void share_wait(share_x2 (&x)[TILES])
{
    if constexpr (TILES == 1)
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x[0]) : "n"(N));
    else
        asm volatile("s_waitcnt lgkmcnt(%2)"
                     : "+v"(x[0]), "+v"(x[1])
                     : "n"(N));
}

template <int TILES, int MODE, bool FMA, int W, int UNR, int AHEAD>
__global__ __launch_bounds__(W *kWave) void spmm_groupshare(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const double *__restrict__ gw,
    const int32_t *__restrict__ grid, const double *__restrict__ gfrac,
    const int64_t *__restrict__ smeta, const int32_t *__restrict__ scol,
    const int32_t *__restrict__ smask, const double *__restrict__ X)
{
    constexpr int G = 8, VEC = 2;
    constexpr int EPW = UNR / W;             // entries a wave sends per step
    constexpr int kEntryBytes = TILES * 1024;
    constexpr int kBufBytes = UNR * kEntryBytes;
    constexpr int kStepsPerBlock = kWave / UNR;   // steps one lane-held block
                                                  // of 64 columns / masks lasts
    constexpr int NW = (UNR * G + kWave - 1) / kWave;
    static_assert(UNR % W == 0 && UNR == 8, "step shape");
    static_assert(AHEAD >= 1 && AHEAD < UNR && AHEAD * TILES <= 15,
                  "LDS reads ahead of the sums");
    static_assert(TILES == 1 || TILES == 2, "K tiles per wave");
    typedef typename I32Vec<G>::type rvec_t;
    typedef typename F64Vec<G>::type fvec_t;
    extern __shared__ __attribute__((aligned(16))) char ring[];   // 2 buffers

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
    // the supergroup's list [0, len) of union entries, this wave's stream of
    // weights: 32-bit positions from here on
    const int64_t s0 = smeta[2 * sg];
    const int len = static_cast<int>(smeta[2 * sg + 2] - s0);
    const int32_t *__restrict__ lcol = scol + s0;
    const int32_t *__restrict__ lmask = smask + s0;
    const double *__restrict__ lw = gw + gmeta[2 * (have ? g : n_groups) + 1];
    const int sh = wave * G;
    // byte offsets from a source row's base; a row's base from its index
    // with one 32 x 32 -> 64 bit product (the host checked both ranges)
    uint32_t xob[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        xob[t] = static_cast<uint32_t>(xoff[t]) * 8u;
    const uint32_t ldx_bytes = static_cast<uint32_t>(p.ldx) * 8u;
    const uint32_t ring_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) char *)ring));

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

    // this wave's pieces of the step whose first union entry is `base`, into
    // ring buffer `buf`; `cols` holds the columns of the 64 entries from
    // `cbase` on, one per lane
    auto send = [&](const int buf, const int32_t cols, const int cbase,
                    const int base) {
#pragma unroll
        for (int i = 0; i < EPW; ++i) {
            const int uu = wave * EPW + i;
            if (base + uu < len) {
                int32_t c =
                    __builtin_amdgcn_readlane(cols, base - cbase + uu);
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
        }
    };

    const int n_steps = (len + UNR - 1) / UNR;
    // columns and masks, 64 entries per block, one per lane (the arrays are
    // padded: always in bounds)
    int32_t colv = lcol[lane];
    int32_t maskv_next = lmask[lane];
    int32_t colv_next = colv;
    // a block of 64 masks -> this wave's member bits of every entry (none
    // behind the list's end) and, in the 8 lanes of a step, the number of
    // bits set in the step
    int32_t bitsv = 0, cntv = 0;
    auto prepare = [&](const int32_t raw, const int block_base) {
        int32_t mine = (raw >> sh) & 0xff;
        mine = block_base + lane < len ? mine : 0;
        int32_t pc = __builtin_popcount(mine);
        pc += __builtin_amdgcn_update_dpp(0, pc, 0xB1, 0xf, 0xf, true);
        pc += __builtin_amdgcn_update_dpp(0, pc, 0x4E, 0xf, 0xf, true);
        pc += __builtin_amdgcn_update_dpp(0, pc, 0x141, 0xf, 0xf, true);
        bitsv = mine;
        cntv = pc;
    };
    double my_w[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q)
        my_w[q] = lw[q * kWave + lane];
    int woff = 0;
    if (n_steps > 0)
        send(0, colv, 0, 0);

    for (int st = 0; st < n_steps; ++st) {
        const int base = st * UNR;
        const int buf = st & 1;
        const int sib = st % kStepsPerBlock;
        share_barrier();
        if (sib == 0)
            prepare(maskv_next, base);
        // step st is in the ring, step st - 1's buffer is free: the next
        // step's pieces leave now and fly while this one is summed
        if (st + 1 < n_steps) {
            const bool wrap = sib + 1 == kStepsPerBlock;
            send(buf ^ 1, wrap ? colv_next : colv,
                 wrap ? base + UNR : base - sib * UNR, base + UNR);
            if (wrap) {
                colv = colv_next;
                maskv_next = lmask[base + UNR + lane];
            }
            // the block of columns after this one, a step before its first
            // piece is sent
            if (sib + 2 == kStepsPerBlock)
                colv_next = lcol[base + 2 * UNR + lane];
        }
        // the member bits of this wave's rows, the number of weights the
        // step takes from the wave's stream
        int bits[UNR];
#pragma unroll
        for (int uu = 0; uu < UNR; ++uu)
            bits[uu] = __builtin_amdgcn_readlane(bitsv, sib * UNR + uu);
        const int cnt = __builtin_amdgcn_readlane(cntv, sib * UNR);
        double w_next[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q)
            w_next[q] = lw[woff + cnt + q * kWave + lane];

        // the step's entries from LDS, AHEAD of the sums
        const uint32_t mine = ring_lds + buf * kBufBytes + lane * 16;
        share_x2 xr[AHEAD + 1][TILES];
        share_static_for(
            std::make_integer_sequence<int, AHEAD>{}, [&](auto d_c) {
                constexpr int d = decltype(d_c)::value;
                share_read<d * kEntryBytes>(xr[d][0], mine);
                if constexpr (TILES == 2)
                    share_read<d * kEntryBytes + 1024>(xr[d][1], mine);
            });
        int idx = 0;   // scalar: next weight of the step
        share_static_for(
            std::make_integer_sequence<int, UNR>{}, [&](auto uu_c) {
                constexpr int uu = decltype(uu_c)::value;
                constexpr int slot = uu % (AHEAD + 1);
                if constexpr (uu + AHEAD < UNR) {
                    constexpr int nx = (uu + AHEAD) % (AHEAD + 1);
                    share_read<(uu + AHEAD) * kEntryBytes>(xr[nx][0], mine);
                    if constexpr (TILES == 2)
                        share_read<(uu + AHEAD) * kEntryBytes + 1024>(
                            xr[nx][1], mine);
                }
                const int b = bits[uu];
                if (b) {
                    // reads issued behind this entry's: those of the entries
                    // uu + 1 ... min(uu + AHEAD, UNR - 1)
                    constexpr int behind =
                        (uu + AHEAD < UNR ? AHEAD : UNR - 1 - uu) * TILES;
                    share_wait<behind, TILES>(xr[slot]);
                    constexpr bool kMasked = MODE == REMAP_MODE_MASKED;
                    double xz[TILES][VEC], vf[TILES][VEC];
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const double x = xr[slot][t][v];
                            if constexpr (kMasked) {
                                // once per entry, reused by every member
                                // row that owns it (spmm_rowgroup.h: kHoist)
                                const bool valid = (x == x);
                                xz[t][v] = valid ? x : 0.0;
                                vf[t][v] = valid ? 1.0 : 0.0;
                                asm volatile(""
                                             : "+v"(xz[t][v]),
                                               "+v"(vf[t][v]));
                            } else {
                                xz[t][v] = x;
                                vf[t][v] = 0.0;
                            }
                        }
#pragma unroll
                    for (int m = 0; m < G; ++m) {
                        if (b & (1 << m)) {
                            double a;
                            if constexpr (NW == 1)
                                a = readlane_f64(my_w[0], idx);
                            else
                                a = idx < kWave
                                        ? readlane_f64(my_w[0], idx)
                                        : readlane_f64(my_w[1],
                                                       idx - kWave);
                            ++idx;
#pragma unroll
                            for (int t = 0; t < TILES; ++t)
#pragma unroll
                                for (int v = 0; v < VEC; ++v) {
                                    acc[m][t][v] = mul_add<FMA>(
                                        a, xz[t][v], acc[m][t][v]);
                                    if constexpr (kMasked)
                                        den[m][t][v] = den_add(
                                            a, vf[t][v], den[m][t][v]);
                                }
                        }
                    }
                }
            });
        woff += cnt;
#pragma unroll
        for (int q = 0; q < NW; ++q)
            my_w[q] = w_next[q];
    }

    if (nmem > 0) {
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
