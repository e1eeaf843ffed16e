// spmm_narrowshare.h -- family 10, the shared form (spmm_groupshare.h) for at
// most 64 columns: ONE 3-D field of up to 64 levels on an entry-rich mapping.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// spmm_groupshare gives a lane two columns per K tile: at 64 columns half of
// every wave idles, and the 8-row groups (one column per lane) are faster --
// each of them pulling its own copy of every source row through L1.  Here a
// lane owns ONE column (64 lanes = the 64 columns of the chunk, 512 bytes per
// source row), the four waves of a workgroup own a 4 x 8 tile of destination
// rows and share ONE union of source rows through the same two-buffer LDS
// ring, and an LDS-DMA instruction carries TWO union entries: lanes 0 - 31
// the 16-byte pieces of one source row, lanes 32 - 63 those of the next (the
// instruction takes an address per lane; the LDS side is linear in the lane).
// A step of 8 entries is ONE global_load_lds_dwordx4 per wave.  Everything
// else -- the lanes holding the list's columns and member bytes, the weights
// through a wave-private LDS slot, the member chain, the sums in ascending
// column order -- is spmm_groupshare's: the same bits.  (Instantiated for the
// frac_b and raw modes.  The masked mode with its per-lane normalisers was
// measured on config 5's mapping and is not: K = 64 2.21 ms against 2.19 of
// the 8-row groups, 34: 2.25 against 2.13 -- profiles/r06_analysis/
// config5_share.md section 9.)
// ---------------------------------------------------------------------------

template <int OFF>
__device__ __forceinline__ void narrow_read(double &x, uint32_t addr)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2"
                 : "=v"(x)
                 : "v"(addr), "n"(OFF));
}

template <int MODE, bool FMA, int AHEAD>
__global__ __launch_bounds__(4 * kWave) void spmm_narrowshare(
    const KParams p, const uint32_t flags,
    const int64_t *__restrict__ gmeta, const double *__restrict__ gw,
    const int32_t *__restrict__ grid, const double *__restrict__ gfrac,
    const int64_t *__restrict__ smeta, const int32_t *__restrict__ scol,
    const int32_t *__restrict__ smask, const double *__restrict__ X)
{
    constexpr int G = 8, VEC = 1, TILES = 1, W = 4, UNR = 8, NBUF = 2;
    constexpr int EPW = UNR / W;             // entries a wave sends per step
    constexpr int A = NBUF - 1;              // steps the DMA runs ahead
    constexpr int kEntryBytes = 512;         // 64 columns of one source row
    constexpr int kBufBytes = UNR * kEntryBytes;
    constexpr int kWSlot = UNR * G * 8;      // a step's weights at most
    constexpr int kWDma = kWSlot / 256;      // ... 256 bytes per instruction
    constexpr int kOps = 1 + kWDma;          // DMAs per wave and step
    constexpr int kSeg = 2 * kWave;          // union entries per segment
    static_assert(EPW == 2, "one DMA instruction = the wave's two entries");
    static_assert(AHEAD >= 1 && AHEAD < UNR, "LDS reads ahead of the sums");
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
    // the sending side: lane -> the wave's first (lanes 0 - 31) or second
    // entry of the step, columns 2 (lane % 32) and the next of the chunk's 64
    // (flat column -> batch, level as tile_offsets cuts them; a pair lies in
    // one batch: the level count is even); columns behind the last one send
    // the row's first piece -- harmless, never summed
    uint64_t xob;
    {
        const uint32_t kf = static_cast<uint32_t>(chunk) * kWave +
                            2u * (lane & 31);
        const bool on = kf < p.K;
        const uint32_t b = on ? kf / p.k_inner : 0u;
        const uint32_t kk = on ? kf - b * p.k_inner : 0u;
        xob = static_cast<uint64_t>(static_cast<int64_t>(b) * p.bsx + kk) *
              8u;
    }
    const bool upper = lane >= 32;
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
            {
                // the wave's two entries of the step in ONE instruction
                int32_t c2[EPW];
#pragma unroll
                for (int i = 0; i < EPW; ++i) {
                    int e = st * UNR + wave * EPW + i;
                    e = e < seg_len ? e : seg_len - 1;   // (same half)
                    c2[i] = __builtin_amdgcn_readlane(col_s,
                                                      e & (kWave - 1));
                    REMAP_DIAG_COL(p, c2[i]);
                }
                const uint32_t c =
                    static_cast<uint32_t>(upper ? c2[1] : c2[0]);
                const char *src = reinterpret_cast<const char *>(X) +
                                  static_cast<uint64_t>(c) * ldx_bytes + xob;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)src,
                    (__attribute__((address_space(3))) void *)(
                        ring + buf * kBufBytes + wave * EPW * kEntryBytes),
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
            const uint32_t mine = ring_lds + buf * kBufBytes + lane * 8;
            double xr[AHEAD + 1];
            share_static_for(
                std::make_integer_sequence<int, AHEAD>{}, [&](auto d_c) {
                    constexpr int d = decltype(d_c)::value;
                    narrow_read<d * kEntryBytes>(xr[d], mine);
                });
            share_wait_w<AHEAD>(my_w);
            int idx = 0;   // scalar: next weight of the step
            share_static_for(
                std::make_integer_sequence<int, UNR>{}, [&](auto uu_c) {
                    constexpr int uu = decltype(uu_c)::value;
                    constexpr int slot = uu % (AHEAD + 1);
                    if constexpr (uu + AHEAD < UNR) {
                        constexpr int nx = (uu + AHEAD) % (AHEAD + 1);
                        narrow_read<(uu + AHEAD) * kEntryBytes>(xr[nx], mine);
                    }
                    // the entry's member byte: tested in place
                    const uint32_t word = uu < 4 ? step_lo : step_hi;
                    constexpr int sb = 8 * (uu & 3);
                    if ((word & (0xffu << sb)) && !REMAP_DIAG_ON(p, 8)) {
                        // reads issued behind this entry's: those of the
                        // entries uu + 1 ... min(uu + AHEAD, UNR - 1)
                        constexpr int behind =
                            uu + AHEAD < UNR ? AHEAD : UNR - 1 - uu;
                        share_wait_w<behind>(xr[slot]);
                        constexpr bool kMasked = MODE == REMAP_MODE_MASKED;
                        double xz[TILES][VEC], vf[TILES][VEC];
#pragma unroll
                        for (int t = 0; t < TILES; ++t)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                const double x = xr[slot];
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
