// spmm_longwave.h -- family 11: one wave per LONG row x 64 columns, the source
// cells a few neighbouring long rows share sliding through LDS in windows.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// For the pole-cap rows of a global bilinear map as ESMF makes it (362-1 442
// entries each, `engine.RemapPlan._split_long_rows`) applied to MANY fields
// (K > 16; family 9 serves the few-field case).  What the other families do
// with such rows at K = 64 (1 440 rows of 360 entries, 1 deg -> 0.5 deg):
//   * family 7 (lanes across 256 rows, column-major entries): a lane walks
//     its row as a chain of entry loads, 6 patches x K / 2 workgroups: 26 us;
//   * family 9 (products staged per (row, columns)): every staged value is
//     used once -- 265 MB through 41 MB of LDS: 87-121 us;
//   * family 5 (all distinct source cells of a patch in LDS at once): 360
//     cells x 512 bytes do not fit.
// Only the SUM of a row has to run in order.  Here a workgroup owns a patch
// of R <= 16 consecutive long rows (one wave each) and one 64-column chunk:
//   * the patch's distinct source cells pass through LDS in WINDOWS of 8 R
//     cells (two buffers; the cells of windows w + 1 and w + 2 are on
//     their way -- one in LDS or about to be written, one in registers --
//     while window w is summed; one barrier per window).  A row's entries are
//     sorted by source cell, so each wave simply walks on through its row
//     while the entries fall into the window -- its running sums stay in
//     registers from window to window: CSR order, the bits of every other
//     family;
//   * (local index, weight) records are read once: all of a row's go to
//     wave-private LDS up front (through the scalar cache every eight
//     records were two trips to L2: 23 us; fetched 64 at a time while the
//     sums ran, a trip to memory the sums waited for: 21 us), from where
//     lane l takes record l % 16 of a batch of 16, the layout DPP
//     `row_newbcast` broadcasts from; the values by
//     `ds_read_b64`, lanes across the columns (conflict-free); per entry:
//     `v_add_u32_dpp` (address), the LDS read, `v_mov_b64_dpp` (weight),
//     one multiply, one add -- sixteen reads ahead of their sums.
// LDS traffic is R x entries x 512 bytes per workgroup (the staging writes
// 1 / R of that), the chain an add per entry.
//
// Measured (1 deg -> 0.5 deg with pole caps, the long rows' launch replayed
// from a hipGraph, R = 6): K = 24 ... 64 26 -> 20 us, K = 128 38 -> 27;
// from K = 256 family 7 is as fast or faster (its lanes share a staged cell
// over 256 rows) and keeps the launch (engine.LONG_WAVE_MAX).  Where the 20
// us go: 3 before the first window is staged (patch -> list -> cells: three
// dependent trips), ~10 of sums -- four VALU instructions per entry and 64
// columns, 1 440 equal chains on 1 024 SIMDs: two per SIMD is the makespan
// -- the rest windows' barriers and the epilogue.
//
// PRECONDITION: `plidx` does not decrease inside a row (true for a plan that
// `remap_patches_build` made from a CSR whose rows are sorted by column --
// every CSR of this library).  `A` holds the long rows only: its row r is
// work slot r; row_order[r] (if given) names the row of Y / frac_b /
// mask_out it writes.
// ---------------------------------------------------------------------------
constexpr int kLongPre = 8;          // source cells a wave stages per window
constexpr int kLongCellBytes = 512;  // one cell's 64 float64 columns in LDS
constexpr int kLongRecordBytes = 12; // per entry in LDS: cell offset, weight

// LDS through a 32-bit address (no generic-pointer arithmetic: `lds + off`
// on the asm's result costs an add of the segment base per read)
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ double lds_read_f64(uint32_t addr)
{
    return *reinterpret_cast<lds_cdouble *>(static_cast<uintptr_t>(addr));
}

// Sixteen records of a batch (one per lane in `off` / `w`, replicated in the
// four 16-lane rows): ALL the LDS reads first, then the products added in
// order.  FULL: every record counts.  Otherwise records [j0, j1) count
// (window borders, row tails): the others are read all the same -- the
// caller points them at the window's first cell -- and their sums dropped
// behind a wave-uniform test.
template <int MODE, bool FMA, bool FULL>
struct LongBatch {
    template <int J>
    static __device__ __forceinline__ void reads(uint32_t off, uint32_t mine,
                                                 double (&x)[16])
    {
        if constexpr (J < 16) {
            x[J] = lds_read_f64(strip_addr<J>(off, mine));
            reads<J + 1>(off, mine, x);
        }
    }
    template <int J>
    static __device__ __forceinline__ void sums(double w,
                                                const double (&x)[16],
                                                int j0, int j1, double &acc,
                                                double &den)
    {
        if constexpr (J < 16) {
            const double a = strip_weight<J>(w);
            double t, d = den;
            if constexpr (MODE == REMAP_MODE_MASKED) {
                const bool valid = (x[J] == x[J]);
                t = mul_add<FMA>(a, valid ? x[J] : 0.0, acc);
                d = den_add(a, valid ? 1.0 : 0.0, den);
            } else {
                t = mul_add<FMA>(a, x[J], acc);
            }
            if (FULL || (J >= j0 && J < j1)) {
                acc = t;
                den = d;
            }
            sums<J + 1>(w, x, j0, j1, acc, den);
        }
    }
    static __device__ __forceinline__ void run(uint32_t off, double w,
                                               uint32_t mine, int j0, int j1,
                                               double &acc, double &den)
    {
        double x[16];
        dpp_settle(off, mine, w);
        reads<0>(off, mine, x);
        // (left alone hipcc keeps three reads ahead of the sums; sixteen
        // measured the same -- the sums are issue-bound -- and cost nothing)
        asm volatile("" ::: "memory");
        sums<0>(w, x, j0, j1, acc, den);
    }
};

template <typename XT, int MODE, bool FMA>
__global__ __launch_bounds__(kPatchBlock) void spmm_longwave(
    const KParams p, const uint32_t flags,
    const int32_t *__restrict__ prow, const double *__restrict__ pval,
    const int32_t *__restrict__ plidx, const int32_t *__restrict__ pptr,
    const int32_t *__restrict__ ucol, const int32_t *__restrict__ row_order,
    const double *__restrict__ frac_b, const int32_t patch_rows,
    const int32_t umax, const int32_t emax, const int64_t n_patches)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    (void)flags;
    (void)umax;
    const int epitch = emax;   // records a wave's LDS area holds (host)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = static_cast<int>(blockDim.x) >> 6;   // waves = patch rows
    const int W = nw * kLongPre;                        // cells per window
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / n_patches;  // chunk-major work list
    const int64_t patch = L - chunk * n_patches;

    int64_t xoff[1], yoff[1];
    bool act[1];
    tile_offsets<1, 1>(p, chunk, lane, xoff, yoff, act);

    const int u0 = pptr[patch];
    const int U = pptr[patch + 1] - u0;
    const int64_t slot0 = p.row_begin + patch * patch_rows;
    const int64_t local0 = patch * patch_rows;  // index into prow
    int nrows = patch_rows;
    if (slot0 + nrows > p.row_end)
        nrows = static_cast<int>(p.row_end - slot0);
    // (waves without a row stage and wait at the barriers like the others)
    const bool has_row = wave < nrows;
    const int rw = has_row ? wave : 0;
    int s = __builtin_amdgcn_readfirstlane(prow[local0 + rw]);
    const int e =
        has_row ? __builtin_amdgcn_readfirstlane(prow[local0 + rw + 1]) : s;
    const int64_t rid = row_order
                            ? static_cast<int64_t>(row_order[slot0 + rw])
                            : slot0 + rw;
    double fb = 0.0;
    if constexpr (MODE == REMAP_MODE_FRACB)
        fb = frac_b[rid];

    const XT *__restrict__ X = static_cast<const XT *>(p.X);
    const int n_win = (U + W - 1) / W;
    // this wave's cells of window w: list positions w * W + q * nw + wave,
    // q = 0 ... 7.  Their source-cell indices for EIGHT windows come with
    // one vector load (lane (w % 8) * 8 + q; clamped to the list: no load
    // behind a divergent branch) and are handed out by v_readlane -- read
    // through the scalar cache they were a trip to L2 per window, in front
    // of the cell loads that depend on them
    int32_t cols_v = 0;
    auto load_cols = [&](int w8) {   // windows w8 ... w8 + 7
        const int c = (w8 + (lane >> 3)) * W + (lane & 7) * nw + wave;
        if (U > 0)
            cols_v = ucol[u0 + (c < U ? c : U - 1)];
    };
    auto fetch_as = [&](int win, XT (&pre)[kLongPre], auto fold_tag) {
        constexpr bool FOLD = decltype(fold_tag)::value;
#pragma unroll
        for (int q = 0; q < kLongPre; ++q) {
            int32_t col =
                __builtin_amdgcn_readlane(cols_v, (win & 7) * 8 + q);
            REMAP_DIAG_COL(p, col);
            const int64_t base = FOLD ? cell_base(p, col)
                                      : static_cast<int64_t>(col) * p.ldx;
            pre[q] = X[base + xoff[0]];
        }
    };
    // (called for win = 0, 1, 2, ... in order)
    auto fetch = [&](int win, XT (&pre)[kLongPre]) {
        if (win >= n_win)
            return;
        if ((win & 7) == 0)
            load_cols(win);
        if (p.src_fold == 0)
            fetch_as(win, pre, std::false_type());
        else
            fetch_as(win, pre, std::true_type());
    };
    // TWO windows ahead of the one being summed: its cells are in registers
    // (pre_a: even windows, pre_b: odd ones) while the window before it
    // waits in the other LDS buffer
    XT pre_a[kLongPre], pre_b[kLongPre];
    fetch(0, pre_a);
    fetch(1, pre_b);

    // the row's records -- read once -- go to wave-private LDS UP FRONT
    // (epitch x 12 bytes per wave: byte offset of the record's cell, u32,
    // then the weights), four coalesced load pairs in flight; fetched while
    // the sums ran they were a trip to memory every 64 records, which the
    // sums then waited for (13 of 21 us).  From there lane l takes record
    // l % 16 of each 16-record batch: the layout DPP `row_newbcast`
    // broadcasts from (spmm_strip.h)
    const int s0 = s;
    const int n = e - s0;
    const int l16 = lane & 15;
    char *ebuf = lds + 2 * W * kLongCellBytes +
                 wave * (epitch * kLongRecordBytes);
    uint32_t *eoff = reinterpret_cast<uint32_t *>(ebuf);
    double *ew = reinterpret_cast<double *>(ebuf + epitch * 4);
    for (int k = 0; k < n; k += 4 * kWave) {
        int32_t li[4];
        double w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // (clamped inside the row)
            const int idx = s0 + k + u * kWave + lane;
            li[u] = plidx[idx < e ? idx : e - 1];
            w[u] = pval[idx < e ? idx : e - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // (the last batch of 16 is read whole: the records behind the
            // row's end repeat its last one -- aligned offsets, never summed)
            const int r = k + u * kWave + lane;
            if (r < ((n + 15) & ~15)) {
                eoff[r] = static_cast<uint32_t>(li[u]) * kLongCellBytes;
                ew[r] = w[u];
            }
        }
    }
    int cur_b = -1;
    uint32_t co = 0u;   // this batch: byte offset of the record's cell ...
    double cw = 0.0;    // ... and its weight, record l % 16 in lane l

    double acc = 0.0, den = 0.0;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) char *)lds));
    auto window = [&](int win, XT (&pre)[kLongPre]) {
        char *buf = lds + (win & 1) * (W * kLongCellBytes);
        // the buffer was last read two windows ago: every wave has passed
        // the previous barrier since
#pragma unroll
        for (int q = 0; q < kLongPre; ++q) {
            const int slot = q * nw + wave;
            if (win * W + slot < U)
                *reinterpret_cast<double *>(buf + slot * kLongCellBytes +
                                            lane * 8) =
                    static_cast<double>(pre[q]);
        }
        lds_barrier();
        fetch(win + 2, pre);
        const int lo = win * W;
        // records of cells below `hi` lie in this window (sorted rows)
        const uint32_t hi = win + 1 < n_win
                                ? static_cast<uint32_t>(lo + W) *
                                      kLongCellBytes
                                : 0xffffffffu;
        // LDS byte address of (cell 0 of the list, this lane's column);
        // + cell * 512 (mod 2^32) lands in the window's buffer
        const uint32_t mine =
            lds_base +
            static_cast<uint32_t>((win & 1) * (W * kLongCellBytes) +
                                  lane * 8) -
            static_cast<uint32_t>(lo) * kLongCellBytes;
        while (s < e) {
            const int rel = s - s0;
            const int b = rel >> 4;
            if (b != cur_b) {
                // (beyond the row's end: whatever the LDS holds, never used)
                co = eoff[16 * b + l16];
                cw = ew[16 * b + l16];
                cur_b = b;
            }
            const int j0 = rel & 15;
            const int m = n - 16 * b < 16 ? n - 16 * b : 16;
            const uint64_t in_window = __builtin_amdgcn_ballot_w64(
                lane >= j0 && lane < m && co < hi);
            const int n_in = __builtin_popcountll(in_window);
            if (n_in == 16)
                LongBatch<MODE, FMA, true>::run(co, cw, mine, 0, 16, acc,
                                                den);
            else if (n_in > 0) {
                // records that do not count (before j0, beyond the window or
                // the row) read the window's first cell: every address stays
                // inside the buffer, their sums are dropped
                const bool counts = l16 >= j0 && l16 < j0 + n_in;
                LongBatch<MODE, FMA, false>::run(
                    counts ? co
                           : static_cast<uint32_t>(lo) * kLongCellBytes,
                    cw, mine, j0, j0 + n_in, acc, den);
            }
            s += n_in;
            if (j0 + n_in < m)
                break;   // the rest of the row lies beyond this window
        }
    };
    for (int win = 0; win < n_win; win += 2) {
        window(win, pre_a);
        if (win + 1 < n_win)
            window(win + 1, pre_b);
    }
    double acc1[1][1] = {{acc}};
    double den1[1][1] = {{den}};
    if (has_row)
        finish_row<1, 1, MODE>(p, rid, fb, act, yoff, acc1, den1);
}
