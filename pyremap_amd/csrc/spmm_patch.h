// spmm_patch.h -- family 5: LDS-staged gather of destination patches.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// patch: LDS-staged gather.  One workgroup owns one PATCH of destination rows
// (a 2-D tile of the destination grid when a row order is installed) x one
// 128-column K-chunk:
//
//   1. gather   every DISTINCT source row the patch references is fetched
//               ONCE, straight into LDS, by LDS-DMA (`global_load_lds_dwordx4`:
//               1 KiB = one row chunk per wave instruction, per-lane source
//               address, no VGPRs, all of a wave's fetches in flight at once);
//   2. barrier  (drains the DMA);
//   3. compute  each wave walks its rows of the patch: (local index, S) pairs
//               come through the SCALAR cache (s_load, 8 entries at a time,
//               no vector-memory instructions), source data from LDS with
//               `ds_read_b128` (lanes across K, sequential sum per lane: the
//               same order and the same bits as the other families), fused
//               epilogue, 16-byte non-temporal stores.
//
// Why: on conservative maps every source row is referenced by nnz/n_a = 3-5
// neighbouring destination rows.  In the register-gather kernels each of
// those references is a separate trip through the CU's vector-memory
// pipeline (texture addresser + L1 miss queue), which is the saturated
// resource; here only distinct rows take that trip and
// the re-touches are LDS reads.  Several workgroups per CU overlap one
// another's gather and compute phases.
//
// Metadata pointers are separate __restrict__ kernel arguments (not members
// of KParams) so hipcc can prove them read-only and use scalar loads.
// ---------------------------------------------------------------------------
constexpr int kPatchBlock = 1024;  // 16 waves
constexpr int kPatchWaves = kPatchBlock / kWave;

// LDS image of one workgroup (row_bytes = 1024, 512 or 256 per staged row
// chunk):
//   [0, umax * row_bytes rounded up to a whole KiB)  the distinct source-row
//                          chunks (a DMA instruction always lands 1 KiB)
//   then                   hdr  16 B[rows]  per row: element offset of its
//                                           Y row (int64), first entry, one
//                                           past its last entry (int32 each)
//                                           -- ONE 16-byte read per row
//                          val  f64[emax]   the patch's weights, slot order
//                          fb   f64[rows]   frac_b of the patch's rows
//                          lidx i32[emax]   their local row indices
__host__ __device__ inline uint32_t patch_lds_bytes(int umax, int emax,
                                                    int rows, int row_bytes)
{
    return ((static_cast<uint32_t>(umax) * row_bytes + 1023u) & ~1023u) +
           static_cast<uint32_t>(emax) * 12u +
           static_cast<uint32_t>(rows) * 24u + 32u;
}

// WC = columns per K-chunk: 128 (two elements per lane) or 64 (one element
// per lane: half the LDS per row, so twice the patch area fits -- for
// mappings whose rows reference many source rows, e.g. 2nd-order
// conservative stencils).  The LDS image always holds float64: WC * 8 bytes
// per staged row chunk.
// DMA = true (float64 X only): LDS-DMA, 16 bytes per lane -- needs 16-byte
// aligned pieces (even strides and level counts).
// DMA = false: every lane loads its WC / 64 elements into registers, converts
// (f32 -> f64: the reference's real lat-lon input is f32,
// tests/test_interpolate.py:492-516; converting the few distinct rows once
// while staging keeps the conversion out of the compute loop, which runs
// once per ENTRY -- config 4: 44 staged rows for 2 300 entries per patch) and
// writes them to LDS.  With WC = 64 that takes any stride and alignment:
// (Time, nCells, 61 levels) on a bilinear map.
template <typename XT, int MODE, bool FMA, int WC, bool DMA,
          int BLOCK = kPatchBlock>
__global__ __launch_bounds__(BLOCK) void spmm_patch(
    const KParams p, const uint32_t flags,
    const int32_t *__restrict__ prow, const double *__restrict__ pval,
    const int32_t *__restrict__ plidx, const int32_t *__restrict__ pptr,
    const int32_t *__restrict__ ucol, const int32_t *__restrict__ row_order,
    const double *__restrict__ frac_b, const int32_t patch_rows,
    const int32_t umax, const int32_t emax, const int64_t n_patches)
{
    static_assert(!DMA || sizeof(XT) == 8, "the DMA moves float64 rows");
    constexpr int kWaves = BLOCK / kWave;   // waves of the workgroup
    constexpr int VEC = WC / kWave;           // elements per lane
    constexpr int kRowBytes = WC * 8;         // staged bytes per row (f64)
    constexpr int kRowsPerDma = 1024 / kRowBytes;  // rows per DMA instruction
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef typename XVec<double, VEC>::type xvec_t;   // LDS holds doubles
    typedef typename XVec<XT, VEC>::type gvec_t;       // X as it is
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    REMAP_CLOCK_BEGIN();
    const int64_t chunk = L / n_patches;  // chunk-major work list
    const int64_t patch = L - chunk * n_patches;

    // compute-phase columns of this lane
    int64_t xoff[1], yoff[1];
    bool act[1];
    tile_offsets<VEC, 1>(p, chunk, lane, xoff, yoff, act);
    // gather-phase columns (DMA): every lane moves 16 B (2 doubles); with
    // 512-byte rows one instruction carries two source rows (lanes 0-31 /
    // 32-63).  Same tile shapes as the compute phase (tile_offsets): whole
    // batches per tile (p.bpc) or the flat column list cut every WC columns.
    int64_t goff;
    {
        constexpr int kLanesPerRow = kWave / kRowsPerDma;
        const uint32_t cc = (lane % kLanesPerRow) * 2;   // column in the tile
        if (p.bpc) {
            const uint32_t n_batch = p.K / p.k_inner;
            const uint32_t bi = cc / p.k_inner;
            const uint32_t kk = cc - bi * p.k_inner;
            const uint32_t bb = static_cast<uint32_t>(chunk) * p.bpc + bi;
            const bool in = bi < p.bpc && bb < n_batch;
            goff = in ? static_cast<int64_t>(bb) * p.bsx + kk : 0;
        } else {
            const uint32_t kf = static_cast<uint32_t>(chunk) * WC + cc;
            const bool in = kf < p.K;
            const uint32_t bb = in ? kf / p.k_inner : 0u;
            const uint32_t kk = in ? kf - bb * p.k_inner : 0u;
            goff = static_cast<int64_t>(bb) * p.bsx + kk;
        }
    }
    const int sub = lane / (kWave / kRowsPerDma);  // which row of the pair

    struct alignas(16) RowHeader {   // what a wave needs to start a row
        int64_t ybase;        // element offset of the row in Y (row * ldy)
        int32_t s, e;         // its entries [s, e) in lds_val / lds_lidx
    };
    RowHeader *lds_hdr =
        reinterpret_cast<RowHeader *>(lds +
                                      ((umax * kRowBytes + 1023) & ~1023));
    double *lds_val = reinterpret_cast<double *>(lds_hdr + patch_rows);
    double *lds_fb = lds_val + emax;
    int32_t *lds_lidx = reinterpret_cast<int32_t *>(lds_fb + patch_rows);

    // 1. gather: distinct source rows by LDS-DMA, the patch's entries by
    //    plain loads (they are contiguous: patch-major CSR).  The phase is a
    //    chain of dependent memory trips with every wave of the workgroup
    //    waiting at the barrier behind it, so loads are issued level by
    //    level: everything addressed by the patch id alone first, then what
    //    those values address, LDS writes last (3 trips instead of 5).
    const int u0 = pptr[patch];
    const int U = pptr[patch + 1] - u0;
    const int64_t slot0 = p.row_begin + patch * patch_rows;
    const int64_t local0 = patch * patch_rows;  // index into prow
    int nrows = patch_rows;
    if (slot0 + nrows > p.row_end)
        nrows = static_cast<int>(p.row_end - slot0);
    const int e0 = prow[local0];
    const int n_e = prow[local0 + nrows] - e0;
    // branch-free (clamped) loads: a load inside a divergent branch makes
    // hipcc wait for it on the spot
    const int tc = tid < nrows ? tid : nrows - 1;
    const int32_t *ro = row_order ? row_order + slot0 : prow + local0;
    const int32_t rid_ld = ro[tc];
    const int32_t my_rid =
        row_order ? rid_ld : static_cast<int32_t>(slot0 + tc);
    const int32_t my_rp = prow[local0 + (tid < nrows ? tid : nrows)];
    const int32_t my_rp1 = prow[local0 + (tid < nrows ? tid + 1 : nrows)];
    constexpr int kPre = 2;  // entry batches held in registers meanwhile
    double ev[kPre];
    int32_t el[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const int t = tid + k * BLOCK;
        ev[k] = 0.0;
        el[k] = 0;
        if (t < n_e) {
            ev[k] = pval[e0 + t];
            el[k] = plidx[e0 + t];
        }
    }
    // before the DMA loop: behind it the wait for my_rid would be vmcnt(0)
    double my_fb = 0.0;
    if constexpr (MODE == REMAP_MODE_FRACB)
        my_fb = frac_b[my_rid];
    const XT *__restrict__ X = static_cast<const XT *>(p.X);
    if constexpr (DMA) {
        for (int j = wave * kRowsPerDma; j < U;
             j += kWaves * kRowsPerDma) {
            // rows of the group that do not exist: fetch the first again
            // (they land behind the list, inside its last KiB: the row
            // region is a whole number of KiB, see patch_lds_bytes)
            const int jj = (j + sub < U) ? j + sub : j;
            int32_t c = ucol[u0 + jj];
            REMAP_DIAG_COL(p, c);
            const XT *g = X + static_cast<int64_t>(c) * p.ldx + goff;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)g,
                (__attribute__((address_space(3))) void *)(lds +
                                                           j * kRowBytes),
                16, 0, 0);
        }
    } else {
        // register staging, four rows in flight per wave; a lane stages
        // the very columns it computes (converted to float64)
        constexpr int kAhead = 4;
        for (int j = wave; j < U; j += kWaves * kAhead) {
            gvec_t v[kAhead];
#pragma unroll
            for (int q = 0; q < kAhead; ++q) {
                const int jq = j + q * kWaves;
                int32_t c = ucol[u0 + (jq < U ? jq : j)];
                REMAP_DIAG_COL(p, c);
                v[q] = load_x<XT, VEC>(X + static_cast<int64_t>(c) * p.ldx +
                                       xoff[0]);
            }
#pragma unroll
            for (int q = 0; q < kAhead; ++q) {
                const int jq = j + q * kWaves;
                if (jq < U) {
                    xvec_t d;
                    if constexpr (VEC == 1) {
                        d = static_cast<double>(v[q]);
                    } else {
                        d[0] = static_cast<double>(v[q][0]);
                        d[1] = static_cast<double>(v[q][1]);
                    }
                    *reinterpret_cast<xvec_t *>(lds + jq * kRowBytes +
                                                lane * (VEC * 8)) = d;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const int t = tid + k * BLOCK;
        if (t < n_e) {
            lds_val[t] = ev[k];
            lds_lidx[t] = el[k];
        }
    }
    for (int t = tid + kPre * BLOCK; t < n_e; t += BLOCK) {
        lds_val[t] = pval[e0 + t];
        lds_lidx[t] = plidx[e0 + t];
    }
    bool plain_row = true;   // frac_b == 1: the row needs no normalising
    if (tid < nrows) {
        RowHeader h;
        h.ybase = static_cast<int64_t>(my_rid) * p.ldy;
        h.s = my_rp - e0;
        h.e = my_rp1 - e0;
        lds_hdr[tid] = h;
        if constexpr (MODE == REMAP_MODE_FRACB) {
            lds_fb[tid] = my_fb;
            plain_row = my_fb == 1.0;
        }
    }
    // 2. everything landed, visible to every wave.  The LDS-DMA rows are
    //    tracked by vmcnt, and a workgroup barrier on gfx950 drains only
    //    lgkmcnt: drain this wave's DMA explicitly (the compiler happens to
    //    emit this wait today, from register dependencies; nothing obliges it
    //    to), so no wave passes the barrier with rows still in flight.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (the barrier also answers: is frac_b 1 on every row of the patch? --
    // bilinear maps: yes everywhere -- then the compute loop below runs
    // without loading, testing or dividing by it)
    const bool all_plain =
        MODE == REMAP_MODE_FRACB ? __syncthreads_and(plain_row) != 0
                                 : (__syncthreads(), false);

    // 3. compute the patch's rows from LDS.  The patch kernel is the one
    //    instruction-issue-bound kernel of the library (DESIGN.md section 6:
    //    its time follows the clock), so the row loop is kept short: one
    //    16-byte header read per row, the Y offset precomputed at gather
    //    time, and no frac_b handling at all when it is 1 on the whole patch.
    const char *mine = lds + lane * (VEC * 8);
    auto rows_loop = [&](auto plain_tag) {
        constexpr bool PLAIN = decltype(plain_tag)::value;
        // the next row's header (and frac_b) is read while this row is being
        // computed: short rows (4 entries of a bilinear map) are a chain of
        // LDS round trips otherwise
        RowHeader nx = {0, 0, 0};
        double nx_fb = 0.0;
        if (wave < nrows) {
            nx = lds_hdr[wave];
            if constexpr (MODE == REMAP_MODE_FRACB && !PLAIN)
                nx_fb = lds_fb[wave];
        }
        for (int r = wave; r < nrows; r += kWaves) {
            const int64_t ybase =
                (static_cast<int64_t>(__builtin_amdgcn_readfirstlane(
                     static_cast<int32_t>(nx.ybase >> 32)))
                 << 32) |
                static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
                    static_cast<int32_t>(nx.ybase)));
            const int s = __builtin_amdgcn_readfirstlane(nx.s);
            const int e = __builtin_amdgcn_readfirstlane(nx.e);
            const double fb_row = nx_fb;
            if (r + kWaves < nrows) {
                nx = lds_hdr[r + kWaves];
                if constexpr (MODE == REMAP_MODE_FRACB && !PLAIN)
                    nx_fb = lds_fb[r + kWaves];
            }
            double acc[1][VEC];
            double den[1][VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[0][v] = 0.0;
                den[0][v] = 0.0;
            }
#pragma unroll 4
            for (int jj = s; jj < e; ++jj) {
                // (index, weight) by LDS broadcast (same address in every
                // lane); measured faster than one coalesced read + v_readlane
                const int32_t li = lds_lidx[jj];
                const double a = lds_val[jj];
                const xvec_t xq = *reinterpret_cast<const xvec_t *>(
                    mine + li * kRowBytes);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double x = elem<xvec_t, VEC>(xq, v);
                    if constexpr (MODE == REMAP_MODE_MASKED) {
                        const bool valid = (x == x);
                        acc[0][v] =
                            mul_add<FMA>(a, valid ? x : 0.0, acc[0][v]);
                        den[0][v] =
                            den_add(a, valid ? 1.0 : 0.0, den[0][v]);
                    } else {
                        acc[0][v] = mul_add<FMA>(a, x, acc[0][v]);
                    }
                }
            }
            if (!act[0])
                continue;
            double y[VEC];
            bool ok[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if constexpr (MODE == REMAP_MODE_RAW || PLAIN) {
                    // (x / 1.0 == x exactly)
                    ok[v] = true;
                    y[v] = acc[0][v];
                } else if constexpr (MODE == REMAP_MODE_FRACB) {
                    ok[v] = fb_row > 0.0;
                    y[v] = !ok[v] ? __builtin_nan("")
                           : (fb_row == 1.0) ? acc[0][v]
                                             : acc[0][v] / fb_row;
                } else {
                    ok[v] = den[0][v] > p.thr;
                    y[v] = ok[v] ? acc[0][v] / den[0][v]
                                 : __builtin_nan("");
                }
            }
            const int64_t o = ybase + yoff[0];
            if (REMAP_DIAG_SKIP_STORE(p, y[0]))
                continue;
            store_y<VEC>(p.Y + o, y);
#ifndef REMAP_STAMPS
            if (p.mask_out) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    p.mask_out[o + v] = ok[v] ? 0 : 1;
            }
#endif
        }
    };
    if (all_plain)
        rows_loop(std::true_type());
    else
        rows_loop(std::false_type());
    REMAP_CLOCK_END();
}
