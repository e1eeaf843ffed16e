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
// SUPERGROUP: a 4 x 4 or 4 x 8 tile of the destination grid (the group tiles
// are walked inside such tiles: remap_groups_build's share_waves).  The
// supergroup has ONE sorted union of source rows (share_col, share_mask: bit
// 8 w + m = member m of wave w owns the entry; remap_share_build).  The list
// is walked in steps of UNR union entries through a ring of NBUF buffers in
// LDS:
//
//   * every wave sends its share of a step's entries, 1 KiB per instruction,
//     straight from global memory into the ring by LDS-DMA
//     (global_load_lds_dwordx4), NBUF - 1 steps ahead of the sums -- each
//     distinct source row enters the CU ONCE per supergroup: 0.16 union
//     entries per entry on config 5 where the 8-row groups have 0.33;
//   * one s_barrier per step: behind it step s is in the ring for every wave
//     and the buffer of step s - 1 is free for step s + NBUF - 1;
//   * every wave reads the step's entries from LDS (ds_read_b128, a few
//     entries ahead of the sums) and adds the ones its own 8 rows own --
//     exactly the inner loop of spmm_rowgroup: member bits from the mask, the
//     wave's weights (group_w of the 8-row schedule: its own contiguous
//     stream) handed over by v_readlane with a running scalar index.  A row
//     adds its own entries in ascending column order: the same bits as every
//     other family.
//
// Inside the step loop EVERY vector-memory instruction is an LDS-DMA and
// every step issues the same number of them -- the step's weights travel the
// same way, into a small wave-private ring -- so that "step s has landed" is
// the immediate of one s_waitcnt vmcnt(N): loads return in order, and a plain
// load issued between the DMAs could only be awaited together with
// everything issued before it (NBUF - 2 steps of lookahead lost).  Entries
// behind the list's end re-send its last entry (an L1 hit).
//
// Columns and masks do not pass through the scalar cache here: a scalar load
// in flight turns every LDS wait into lgkmcnt(0).  They are fetched once per
// segment of 128 union entries (nearly every list is one segment), one per
// lane, cut down to this wave's 8 member bits, and handed out by v_readlane;
// the weights each step takes are counted on the vector side (DPP adds) and
// summed up once per segment.  (The first build extracted bits and counts
// entry by entry on the scalar unit: 100 SALU instructions per step, 5.7e9
// per launch against the row-group kernel's 3.5e9, and an on-chip floor of
// 17.7 ms where that kernel has 14.6 -- profiles/r06_analysis.)
// ---------------------------------------------------------------------------

// compile-time loop: the body sees its index as a constant (the offsets of
// the ds_read_b128 below are instruction immediates)
template <int... I, typename F>
__device__ __forceinline__ void share_static_for(
    std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}

// The DMAs of this wave up to the N last ones have landed, every LDS read of
// the last step is done; then the workgroup's barrier.  (Not __syncthreads():
// that drains vmcnt altogether.)
template <int N>
__device__ __forceinline__ void share_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier"
                 :
                 : "n"(N)
                 : "memory");
}

// The same with the read of the step's weights -- the wave's own slot, landed
// once ITS DMAs have -- issued in front of the barrier: its trip to LDS runs
// while the other waves arrive.
#ifdef REMAP_DIAG
// (ablation: the waits and the read without the barrier)
template <int N>
__device__ __forceinline__ void share_nobarrier_w(double &w, uint32_t addr)
{
    asm volatile("s_waitcnt vmcnt(%2) lgkmcnt(0)\n\t"
                 "ds_read_b64 %0, %1"
                 : "=v"(w)
                 : "v"(addr), "n"(N)
                 : "memory");
}
#endif

template <int N>
__device__ __forceinline__ void share_barrier_w(double &w, uint32_t addr)
{
    asm volatile("s_waitcnt vmcnt(%2) lgkmcnt(0)\n\t"
                 "ds_read_b64 %0, %1\n\ts_barrier"
                 : "=v"(w)
                 : "v"(addr), "n"(N)
                 : "memory");
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

// the step's weights: lane j's is weight j of the wave's slot
__device__ __forceinline__ void share_read_w(double &w, uint32_t addr)
{
    asm volatile("ds_read_b64 %0, %1" : "=v"(w) : "v"(addr));
}

template <int N>
__device__ __forceinline__ void share_wait_w(double &w)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w) : "n"(N));
}

// Member bytes of the UNR entries of a step, packed into the step's first
// lane: lo = entries 0 - 3 (byte j = entry j), hi = entries 4 - 7.
template <int UNR>
__device__ __forceinline__ void share_pack_step(int32_t mine, int32_t &lo,
                                                int32_t &hi)
{
    int32_t t = mine | (__builtin_amdgcn_update_dpp(0, mine, 0x101, 0xf, 0xf,
                                                    true)
                        << 8);
    t |= __builtin_amdgcn_update_dpp(0, t, 0x102, 0xf, 0xf, true) << 16;
    lo = t;
    hi = UNR == 8 ? __builtin_amdgcn_update_dpp(0, t, 0x104, 0xf, 0xf, true)
                  : 0;
}

// v = w behind a SCALAR branch (left to itself hipcc turns `half ? a : b`
// into s_cmp, s_cselect, v_cndmask in every step of the loop)
__device__ __forceinline__ void share_switch(int32_t &v, int32_t w)
{
    asm volatile("v_mov_b32 %0, %1" : "+v"(v) : "v"(w));
}

template <int TILES, int MODE, bool FMA, int W, int UNR, int NBUF, int AHEAD>
__global__ __launch_bounds__(W *kWave) void spmm_groupshare(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const double *__restrict__ gw,
    const int32_t *__restrict__ grid, const double *__restrict__ gfrac,
    const int64_t *__restrict__ smeta, const int32_t *__restrict__ scol,
    const int32_t *__restrict__ smask, const double *__restrict__ X)
{
    constexpr int G = 8, VEC = 2;
    constexpr int EPW = UNR / W;             // entries a wave sends per step
    constexpr int A = NBUF - 1;              // steps the DMA runs ahead
    constexpr int kEntryBytes = TILES * 1024;
    constexpr int kBufBytes = UNR * kEntryBytes;
    constexpr int kWSlot = UNR * G * 8;      // a step's weights at most
    constexpr int kWDma = kWSlot / 256;      // ... 256 bytes per instruction
    constexpr int kOps = EPW * TILES + kWDma;    // DMAs per wave and step
    constexpr int kSeg = 2 * kWave;          // union entries per segment
    static_assert(UNR % W == 0 && (UNR == 4 || UNR == 8), "step shape");
    static_assert(NBUF >= 2 && NBUF <= 4 && (A - 1) * kOps <= 63, "ring");
    static_assert(AHEAD >= 1 && AHEAD < UNR && AHEAD * TILES <= 15,
                  "LDS reads ahead of the sums");
    static_assert(TILES == 1 || TILES == 2, "K tiles per wave");
    typedef typename I32Vec<G>::type rvec_t;
    typedef typename F64Vec<G>::type fvec_t;
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
    // the supergroup's list [0, len) of union entries, this wave's stream of
    // weights: 32-bit positions from here on
    const int64_t s0 = smeta[2 * sg];
    const int len = static_cast<int>(smeta[2 * sg + 2] - s0);
    const int32_t *__restrict__ lcol = scol + s0;
    const int32_t *__restrict__ lmask = smask + s0;
    const double *__restrict__ lw = gw + gmeta[2 * (have ? g : n_groups) + 1];
    const int sh = wave * G;
    // byte offsets from a source row's base -- 64 bits: the batches of a
    // (Time, nCells, nVertLevels) field on a 3.7 M-cell mesh are 1.9 GB
    // apart, and the DMA takes a flat address per lane anyway; a row's base
    // from its index with one 32 x 32 -> 64 bit product (the host checked
    // that range)
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

    int seg_w = 0;   // weights of this wave's stream the earlier segments took
    for (int seg0 = 0; seg0 < len; seg0 += kSeg) {
        const int seg_len = (len - seg0) < kSeg ? len - seg0 : kSeg;
        const int seg_steps = (seg_len + UNR - 1) / UNR;
        if (seg0 > 0)   // the ring of the segment before is read to the end
            share_barrier<0>();
        // columns and masks of the segment, one entry per lane and block (the
        // arrays are padded: always in bounds); the masks cut down to this
        // wave's member bits (none behind the list's end); in the lanes of a
        // step the number of bits set in the step
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
            if constexpr (UNR == 8)
                pc += __builtin_amdgcn_update_dpp(0, pc, 0x141, 0xf, 0xf,
                                                  true);
            // a step's member bytes side by side in the step's first lane
            // (row_shl: lane i reads lane i + n of its row of 16): entries
            // 0 - 3 of the step in bitsv, 4 - 7 in bitsh -- two v_readlane
            // per step instead of eight
            share_pack_step<UNR>(mine, bitsv[b], bitsh[b]);
            cntv[b] = pc;
        }
        // (the loads above are awaited HERE, in straight-line code: met
        // first behind a branch, hipcc's wait-count pass no longer knows
        // whether they are still in flight and puts `s_waitcnt vmcnt(0)` in
        // front of every send of the pipeline's fill -- each of them then
        // waits for the one before to land)
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
        // the half of the segment (64 entries: one register of columns, two
        // of member bytes) the sending side / the summing side is in
        int32_t col_s = colv[0], bits_lo = bitsv[0], bits_hi = bitsh[0];
        auto send = [&](const int st) {
            const int buf = st % NBUF;
            if (st * UNR == kWave)
                share_switch(col_s, colv[1]);
            if (REMAP_DIAG_ON(p, 16))
                return;
#pragma unroll
            for (int i = 0; i < EPW; ++i) {
                const int uu = wave * EPW + i;
                int e = st * UNR + uu;
                e = e < seg_len ? e : seg_len - 1;   // (same step, same half)
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

        // the pipeline fills: steps 0 ... A - 1 leave
#pragma unroll
        for (int st = 0; st < A; ++st)
            if (st < seg_steps)
                send(st);

        for (int st = 0; st < seg_steps; ++st) {
            const int buf = st % NBUF;
            // step st has landed in this wave's eyes when at most the DMAs
            // of the A - 1 steps behind it are in flight (the list's last
            // steps: fewer are, everything is awaited)
            double my_w;
            const uint32_t my_w_lds =
                wring_lds + (buf * W + wave) * kWSlot + lane * 8;
#ifdef REMAP_DIAG
            if (REMAP_DIAG_ON(p, 4))
                share_nobarrier_w<0>(my_w, my_w_lds);
            else
#endif
            if (st + A - 1 < seg_steps)
                share_barrier_w<(A - 1) * kOps>(my_w, my_w_lds);
            else
                share_barrier_w<0>(my_w, my_w_lds);
            // ... and in everybody's behind the barrier, and the buffer of
            // step st - 1 is free: step st + A leaves
            if (st + A < seg_steps)
                send(st + A);
            const int e0 = st * UNR;
            if (e0 == kWave) {
                share_switch(bits_lo, bitsv[1]);
                share_switch(bits_hi, bitsh[1]);
            }
            const uint32_t step_lo = static_cast<uint32_t>(
                __builtin_amdgcn_readlane(bits_lo, e0 & (kWave - 1)));
            const uint32_t step_hi =
                UNR == 8 ? static_cast<uint32_t>(__builtin_amdgcn_readlane(
                               bits_hi, e0 & (kWave - 1)))
                         : 0u;

            // the step's entries from LDS, AHEAD of the sums (its weights
            // were asked for in front of the barrier)
            const uint32_t mine = ring_lds + buf * kBufBytes + lane * 16;
            share_x2 xr[AHEAD + 1][TILES];
            share_static_for(
                std::make_integer_sequence<int, AHEAD>{}, [&](auto d_c) {
                    constexpr int d = decltype(d_c)::value;
                    share_read<d * kEntryBytes>(xr[d][0], mine);
                    if constexpr (TILES == 2)
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
                        if constexpr (TILES == 2)
                            share_read<(uu + AHEAD) * kEntryBytes + 1024>(
                                xr[nx][1], mine);
                    }
                    // the entry's member byte: tested in place
                    const uint32_t word = uu < 4 ? step_lo : step_hi;
                    constexpr int sb = 8 * (uu & 3);
                    if ((word & (0xffu << sb)) && !REMAP_DIAG_ON(p, 8)) {
                        // reads issued behind this entry's: those of the
                        // entries uu + 1 ... min(uu + AHEAD, UNR - 1)
                        constexpr int behind =
                            (uu + AHEAD < UNR ? AHEAD : UNR - 1 - uu) *
                            TILES;
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
                                    // row that owns it (spmm_rowgroup.h)
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
                            if (word & (1u << (sb + m))) {
                                const double a = readlane_f64(my_w, idx);
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
        }
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
