// spmm_strip.h -- family 8: an LDS ring sliding along a strip of the
// destination grid, fed by loader waves, drained by compute waves.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// For entry-rich maps (2nd-order conservative stencils: BASELINE config 5,
// 12-30 entries per row).  There the wave-per-row(-group) kernels are bound
// by on-chip traffic, not by HBM: every group of 8 destination rows pulls the
// union of its rows' source rows through the CU's L1 again -- 211 GB of L2 ->
// L1 fills for 30 GB of X.  Here every source-row piece
// enters a CU ONCE per strip:
//
//   * the destination grid is cut into strips of R grid rows, a strip into
//     segments; one workgroup owns one (segment, 64-column K-chunk) and walks
//     it in STEPS of W grid columns (R x W destination rows per step);
//   * everything a step reads ARRIVES in LDS by LDS-DMA
//     (`global_load_lds_dwordx4`), issued DEPTH steps ahead by DEPTH loader
//     waves that do nothing else: the 512-byte pieces (64 columns) of the
//     source rows that enter, two per instruction, into a ring of slots -- a
//     piece stays while consecutive steps use it, so the overlap of
//     neighbouring steps' stencils is never fetched again, only the halo
//     above and below the strip is -- and the step's META block (row
//     headers, then one 16-byte record per entry: LDS offset of its piece,
//     weight) into one of DEPTH + 1 meta slots;
//   * the compute waves (up to 16 - DEPTH) share the step's R W rows.  They issue no
//     global load at all (a wave's loads and stores share ONE in-order
//     counter, vmcnt: a load behind Y stores would wait for them): entry
//     records come from LDS sixteen at a time, lane l holding record l % 16,
//     and reach all lanes by DPP (`row_newbcast`: lane j of every 16-lane
//     row; the four rows hold the same records) -- `v_add_u32_dpp` forms the
//     piece address, `v_mov_b64_dpp` hands the weight over; X values by
//     `ds_read_b64` (64 lanes x 8 bytes of one piece: conflict-free); the
//     sum runs in CSR order in every lane -- the bits of scipy's
//     csr_matvecs; the fused epilogue stores 512 contiguous bytes of Y;
//   * one s_barrier per step; a loader wave waits for ITS arrivals
//     (`s_waitcnt vmcnt(0)`) right before the barrier that precedes the step
//     that reads them.  The compute waves never wait for their Y stores.
//
// The schedule is precomputed per mapping: remap_strips,
// pyremap_amd/strips.py.  Slots are handed out in pairs (one DMA instruction
// = two pieces) from a free list: a pair is reused once both its pieces'
// last readers are done, so the LDS holds little more than what is alive --
// two workgroups per CU.  Rows are padded to a multiple of four records with
// weight +0.0 on a slot of zeros: acc + (+0.0 * +0.0) == acc bit for bit for
// every acc such a sum can hold (it starts at +0.0, hence never is -0.0), so
// the inner loop runs four records at a time without a test.
// ---------------------------------------------------------------------------
constexpr int kStripRowBytes = 512;   // one piece: 64 float64 columns
constexpr int kStripQuant = 4;        // rows hold a multiple of 4 records
// s_waitcnt vmcnt(0) with expcnt / lgkmcnt left alone (gfx9 encoding)
constexpr int kStripWaitVm0 = 0x0F70;

// One barrier per step.  Not __syncthreads(): that also drains vmcnt, and
// the compute waves' Y stores are to stay in flight.  LDS reads are drained
// (lgkmcnt); the memory clobber keeps the compiler from moving LDS accesses
// across it.
__device__ __forceinline__ void strip_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The DPP read hazard: a DPP instruction must not read a VGPR a VALU
// instruction wrote less than 2 wait states before it, and LLVM's hazard
// recogniser does not look into inline asm (which is how the DPP operations
// below are issued).  Every batch passes its DPP operands through this once:
// whatever wrote them, two wait states lie between that and the first DPP
// read.  (A `v_cndmask` put in front of `strip_addr` in round 5 did produce
// wrong sums on the device before this; tools/dpp_hazard_scan.py checks
// every DPP site of the built library.)
__device__ __forceinline__ void dpp_settle(uint32_t &off, uint32_t &lane8,
                                           double &w)
{
    asm volatile("s_nop 1" : "+v"(off), "+v"(lane8), "+v"(w));
}

// lane j of every 16-lane row, to all lanes of that row (DPP row_newbcast)
template <int J>
__device__ __forceinline__ uint32_t strip_addr(uint32_t off, uint32_t lane8)
{
    uint32_t r;
    asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "=v"(r)
        : "v"(off), "v"(lane8), "n"(J));
    return r;
}

template <int J>
__device__ __forceinline__ double strip_weight(double w)
{
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
        : "=v"(r)
        : "v"(w), "n"(J));
    return r;
}

// Q x 4 records of a batch of 16 (one per lane in `off` / `w`, replicated in
// the four 16-lane rows): ALL the LDS reads first, then the products added
// in order.  No test: a row holds a multiple of four records (pads: weight
// +0.0 on the slot of zeros).
template <int MODE, bool FMA, int Q>
struct StripBatch {
    template <int J>
    static __device__ __forceinline__ void reads(uint32_t off,
                                                 const char *lds,
                                                 uint32_t lane8,
                                                 double (&x)[16])
    {
        if constexpr (J < 4 * Q) {
            x[J] = *reinterpret_cast<const double *>(
                lds + strip_addr<J>(off, lane8));
            reads<J + 1>(off, lds, lane8, x);
        }
    }
    template <int J>
    static __device__ __forceinline__ void sums(double w,
                                                const double (&x)[16],
                                                double &acc, double &den)
    {
        if constexpr (J < 4 * Q) {
            const double a = strip_weight<J>(w);
            if constexpr (MODE == REMAP_MODE_MASKED) {
                const bool valid = (x[J] == x[J]);
                acc = mul_add<FMA>(a, valid ? x[J] : 0.0, acc);
                den = den_add(a, valid ? 1.0 : 0.0, den);
            } else {
                acc = mul_add<FMA>(a, x[J], acc);
            }
            sums<J + 1>(w, x, acc, den);
        }
    }
    static __device__ __forceinline__ void run(uint32_t off, double w,
                                               const char *lds,
                                               uint32_t lane8, double &acc,
                                               double &den)
    {
        double x[16];
        dpp_settle(off, lane8, w);
        reads<0>(off, lds, lane8, x);
        sums<0>(w, x, acc, den);
    }
};

template <int MODE, bool FMA, int DEPTH>
__global__ __launch_bounds__(1024)
void spmm_strip(
    const KParams p, const int32_t *__restrict__ unit_steps,
    const int32_t *__restrict__ arr_ptr, const int32_t *__restrict__ arr_src,
    const int32_t *__restrict__ arr_slot,
    const int64_t *__restrict__ meta_ptr, const char *__restrict__ meta,
    const int32_t spu, const int32_t rpw, const int32_t cap,
    const int32_t meta_slot_bytes, const int32_t nw, const int64_t n_chunks)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    // unit-major: the K-chunks of one (strip, segment) run side by side on
    // one XCD -- its schedule is read from HBM once, and the 64-column
    // pieces of a source row are asked for together
    const int64_t unit = L / n_chunks;
    const int64_t chunk = L - unit * n_chunks;
    const int64_t g0 = unit * spu;
    const int T = unit_steps[unit];
    const int64_t col0 = chunk * (kStripRowBytes / 8);
    // LDS: the piece slots, 1 KiB holding the slot of zeros, the meta slots
    char *const zeros = lds + cap * kStripRowBytes;
    char *const meta_lds = zeros + 2 * kStripRowBytes;

    if (wave >= nw) {
        // ---------------- loader wave j: arrivals of steps s = j (mod DEPTH)
        const int j = wave - nw;
        const double *__restrict__ X = static_cast<const double *>(p.X);
        // this lane's 16 bytes of a piece: the two rows of a pair on lanes
        // 0-31 / 32-63
        const int half = lane >> 5;
        const int64_t piece = col0 + (lane & 31) * 2;
        const bool in_k = piece < static_cast<int64_t>(p.K);
        // The lists of a step -- its arrival pairs and their slots -- are
        // read one issue AHEAD into registers (lane i: source rows of pair
        // i / 2's halves, slot of pair i): by the time a step's arrivals
        // are issued nothing is waited for.  (Read one by one through the
        // scalar cache they cost a memory trip per pair: 3 us per step.)
        int lo = 0, hi = 0;
        int64_t m0 = 0, m1 = 0;
        int32_t my_src = -1, my_src2 = -1, my_slot = 0;
        auto fetch_lists = [&](int s) {
            lo = arr_ptr[g0 + s];
            hi = arr_ptr[g0 + s + 1];
            m0 = meta_ptr[g0 + s];
            m1 = meta_ptr[g0 + s + 1];
            const int np = hi - lo;      // at most 64 (checked by the host)
            my_src = lane < 2 * np ? arr_src[2 * lo + lane] : -1;
            my_src2 = lane + 64 < 2 * np ? arr_src[2 * lo + 64 + lane] : -1;
            my_slot = lane < np ? arr_slot[lo + lane] : 0;
        };
        auto issue = [&](int s) {
            // the step's meta block, 1 KiB per instruction (reading past
            // its end is harmless: the array is padded, the slot sized for
            // the largest block rounded up)
            char *mslot = meta_lds + (s % (DEPTH + 1)) * meta_slot_bytes;
            for (int64_t b = m0 * 16; b < m1 * 16; b += 1024) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(
                        meta + b + lane * 16),
                    (__attribute__((address_space(3))) void *)mslot, 16, 0,
                    0);
                mslot += 1024;
            }
            // (no memory access in these loops: a load here would make
            // hipcc wait for it -- and for every DMA before it -- per pair)
            const int np = hi - lo;
            auto one = [&](int32_t list, int q, int k) {
                const int32_t src =
                    __builtin_amdgcn_ds_bpermute((2 * k + half) * 4, list);
                const int32_t pair = __builtin_amdgcn_readlane(my_slot, q);
                if (src >= 0 && in_k) {
                    const double *g =
                        X + static_cast<int64_t>(src) * p.ldx + piece;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)g,
                        (__attribute__((address_space(3))) void *)(
                            lds + pair * (2 * kStripRowBytes)),
                        16, 0, 0);
                }
            };
            const int n1 = np < 32 ? np : 32;
            for (int q = 0; q < n1; ++q)
                one(my_src, q, q);
            for (int q = 32; q < np; ++q)
                one(my_src2, q, q - 32);
        };
        // Barrier number b (0 .. T) precedes step b; the arrivals of step s
        // must have landed before barrier s and may be issued once barrier
        // s - DEPTH is passed (step s - DEPTH running: the slots they take
        // were last read by earlier steps).  The wait is the builtin, not
        // inline asm: hipcc then KNOWS the lists have arrived and puts no
        // wait of its own between the DMAs of the next issue.
        int passed = 0;
        if (j < T) {
            fetch_lists(j);
            __builtin_amdgcn_s_waitcnt(kStripWaitVm0);
            issue(j);
            if (j + DEPTH < T)
                fetch_lists(j + DEPTH);
        }
        for (int s = j; s < T; s += DEPTH) {
            for (; passed < s; ++passed)
                strip_barrier();
            __builtin_amdgcn_s_waitcnt(kStripWaitVm0);   // step s is here
            strip_barrier();
            ++passed;
            if (s + DEPTH < T) {
                issue(s + DEPTH);
                if (s + 2 * DEPTH < T)
                    fetch_lists(s + 2 * DEPTH);
            }
        }
        for (; passed < T + 1; ++passed)
            strip_barrier();
        return;
    }

    // -------------------- compute wave: rpw rows of every step, LDS only
    const int64_t col = col0 + lane;
    bool act[1] = {col < static_cast<int64_t>(p.K)};
    int64_t yoff[1] = {act[0] ? col : 0};
    const uint32_t lane8 = lane * 8;
    const int l16 = lane & 15;
    const int rps = nw * rpw;
    if (wave == 0)
        *reinterpret_cast<double *>(zeros + lane8) = 0.0;
    strip_barrier();          // step 0 has arrived, the zeros are written
    for (int t = 0; t < T; ++t) {
        const char *mb = meta_lds + (t % (DEPTH + 1)) * meta_slot_bytes;
        const char *recs = mb + rps * 32;
        for (int i = 0; i < rpw; ++i) {
            // header: {rid, first record, records, -}, {frac_b, -}
            const char *hp = mb + (wave * rpw + i) * 32;
            const int4 h = *reinterpret_cast<const int4 *>(hp);
            const int32_t rid = __builtin_amdgcn_readfirstlane(h.x);
            if (rid < 0)
                continue;
            const int e0 = __builtin_amdgcn_readfirstlane(h.y);
            const int n = __builtin_amdgcn_readfirstlane(h.z);
            double fb = 0.0;
            if constexpr (MODE == REMAP_MODE_FRACB)
                fb = *reinterpret_cast<const double *>(hp + 16);
            double acc = 0.0, den = 0.0;
            // the batch's records are fetched one batch ahead
            int4 rec = *reinterpret_cast<const int4 *>(recs +
                                                       (e0 + l16) * 16);
            for (int b = 0; b < n; b += 16) {
                const int m = n - b;      // records left (a multiple of 4)
                const uint32_t off = static_cast<uint32_t>(rec.x);
                const double w = __hiloint2double(rec.w, rec.z);
                if (m > 16)
                    rec = *reinterpret_cast<const int4 *>(
                        recs + (e0 + b + 16 + l16) * 16);
                if (m >= 16)
                    StripBatch<MODE, FMA, 4>::run(off, w, lds, lane8, acc,
                                                  den);
                else if (m == 12)
                    StripBatch<MODE, FMA, 3>::run(off, w, lds, lane8, acc,
                                                  den);
                else if (m == 8)
                    StripBatch<MODE, FMA, 2>::run(off, w, lds, lane8, acc,
                                                  den);
                else
                    StripBatch<MODE, FMA, 1>::run(off, w, lds, lane8, acc,
                                                  den);
            }
            const double acc1[1][1] = {{acc}};
            const double den1[1][1] = {{den}};
            finish_row<1, 1, MODE>(p, rid, fb, act, yoff, acc1, den1);
        }
        // every LDS read of this step has been consumed (the sums depend on
        // them); the stores are left in flight
        strip_barrier();
    }
}
