// spmm_patchcell.h -- family 7: LDS-staged patches, lanes ACROSS destination rows.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// patchcell: the (Time, nCells) layouts of family 4 (spmm_rowcell) through
// the LDS patch plan of family 5.  In such a field the K values of one source
// cell lie n_a elements apart, so every X access is an 8-byte gather from a
// 128-byte line.  Family 4 takes one such trip through the CU's L1 per
// (entry, field) -- 3-5 x the distinct cells, and on a real mesh numbering
// the 64 lanes of an instruction hit ~64 different lines: the L2 -> L1 fill
// rate (one line for 8 useful bytes) is the bound, 0.21 of the roofline on
// EC30to60 numbered as MPAS numbers its cells, 0.33 numbered along the raster,
// 0.06 numbered at random.
//
// Here one workgroup owns one PATCH of destination rows (a tile of the
// destination grid) x TT fields:
//   1. stage    every DISTINCT source cell of the patch, for each of the TT
//               fields, is loaded ONCE: lanes run across the patch's SORTED
//               list of source cells, so cells with neighbouring ids -- the
//               runs of 3-9 a mesh generator leaves -- share a line fetch
//               inside one instruction.  LDS layout [field][cell].
//   2. barrier
//   3. compute  lane = destination row (slot) of the patch; its (local
//               index, S) pairs are lane-private, X comes from LDS
//               (ds_read_b64, 512 B per instruction instead of 64 line
//               fills), sums in CSR order, multiply then add: same bits as
//               every other family.  Consecutive slots of a tile row are
//               consecutive destination rows: Y stores are coalesced.
// f32 fields are converted while staging (LDS holds doubles).
// ---------------------------------------------------------------------------
constexpr int kCellBlock = 256;

// LAYOUT of the patch's entries -- 0: row after row (patch_rowptr); 1, for
// patches of LONG rows: column-major (remap_apply_args.patch_ell_base), the
// lanes' j-th entries are one coalesced load.
template <typename XT, int MODE, bool FMA, int TT, int LAYOUT = 0>
__global__ __launch_bounds__(kCellBlock) void spmm_patchcell(
    const KParams p, const uint32_t flags,
    const int32_t *__restrict__ prow, const double *__restrict__ pval,
    const int32_t *__restrict__ plidx, const int32_t *__restrict__ pptr,
    const int32_t *__restrict__ ucol, const int32_t *__restrict__ row_order,
    const double *__restrict__ frac_b, const int32_t patch_rows,
    const int32_t upitch, const int64_t n_patches,
    const int64_t *__restrict__ ell_base)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    double *xs = reinterpret_cast<double *>(lds);   // [TT][upitch]
    const int tid = threadIdx.x;
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / n_patches;  // chunk-major work list
    const int64_t patch = L - chunk * n_patches;

    // wave-uniform element offsets of this chunk's TT flat columns
    int64_t xo[TT], yo[TT];
    bool act[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const uint32_t kf = static_cast<uint32_t>(chunk) * TT + t;
        act[t] = kf < p.K;
        const uint32_t b = act[t] ? kf / p.k_inner : 0u;
        const uint32_t k = act[t] ? kf - b * p.k_inner : 0u;
        xo[t] = static_cast<int64_t>(b) * p.bsx + k;
        yo[t] = static_cast<int64_t>(b) * p.bsy + k;
    }
    const XT *__restrict__ X = static_cast<const XT *>(p.X);

    // 1. stage the patch's distinct source cells, TT fields each
    const int u0 = pptr[patch];
    const int U = pptr[patch + 1] - u0;
    for (int j = tid; j < U; j += kCellBlock) {
        const int64_t c = cell_base(p, ucol[u0 + j]);
        XT v[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t)
            v[t] = X[c + xo[t]];
#pragma unroll
        for (int t = 0; t < TT; ++t)
            xs[t * upitch + j] = static_cast<double>(v[t]);
    }
    __syncthreads();

    // 2. the patch's rows, one lane each
    const int64_t slot0 = p.row_begin + patch * patch_rows;
    const int64_t local0 = patch * patch_rows;  // index into prow
    int nrows = patch_rows;
    if (slot0 + nrows > p.row_end)
        nrows = static_cast<int>(p.row_end - slot0);
    (void)flags;
    for (int r = tid; r < nrows; r += kCellBlock) {
        const int64_t i = row_order ? (int64_t)row_order[slot0 + r]
                                    : slot0 + r;
        const int s = prow[local0 + r];
        const int e = prow[local0 + r + 1];
        double acc[TT], den[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            acc[t] = 0.0;
            den[t] = 0.0;
        }
        auto add_entry = [&](const int32_t li, const double a) {
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const double x = xs[t * upitch + li];
                if constexpr (MODE == REMAP_MODE_MASKED) {
                    const bool valid = (x == x);
                    acc[t] = mul_add<FMA>(a, valid ? x : 0.0, acc[t]);
                    den[t] = den_add(a, valid ? 1.0 : 0.0, den[t]);
                } else {
                    acc[t] = mul_add<FMA>(a, x, acc[t]);
                }
            }
        };
        if constexpr (LAYOUT == 1) {
            // this lane's j-th entry sits patch_rows entries behind its
            // (j - 1)-th, next to the other lanes' j-th entries: coalesced
            // U entries' (index, weight) loads in flight at once: the row is
            // one dependent chain and every trip to memory is paid in full,
            // ~0.3 us -- 4 entries per trip: 29 us for a 362-entry row
            // whatever TT; 32 per trip (TT = 1): 15 us.  (With 4 fields per
            // lane 8 per trip ran slower than 4: registers.)
            constexpr int U = TT <= 1 ? 32 : TT == 2 ? 16 : 4;
            const int64_t first = ell_base[patch] + r;
            const int n_e = e - s;
            int j = 0;
            for (; j + U <= n_e; j += U) {
                int32_t li[U];
                double a[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t jj = first + (int64_t)(j + u) * patch_rows;
                    li[u] = plidx[jj];
                    a[u] = pval[jj];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    add_entry(li[u], a[u]);
            }
            for (; j < n_e; ++j) {
                const int64_t jj = first + (int64_t)j * patch_rows;
                add_entry(plidx[jj], pval[jj]);
            }
        } else {
            for (int jj = s; jj < e; ++jj)
                add_entry(plidx[jj], pval[jj]);
        }
        double fb = 0.0;
        if constexpr (MODE == REMAP_MODE_FRACB)
            fb = frac_b[i];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (!act[t])
                continue;
            bool ok = true;
            double y = acc[t];
            if constexpr (MODE == REMAP_MODE_FRACB) {
                ok = fb > 0.0;
                y = !ok ? __builtin_nan("")
                    : (fb == 1.0) ? acc[t] : acc[t] / fb;
            } else if constexpr (MODE == REMAP_MODE_MASKED) {
                ok = den[t] > p.thr;
                y = ok ? acc[t] / den[t] : __builtin_nan("");
            }
            const int64_t o = i * p.ldy + yo[t];
            __builtin_nontemporal_store(y, p.Y + o);
#ifndef REMAP_STAMPS   // (there mask_out is the stamps buffer)
            if (p.mask_out)
                p.mask_out[o] = ok ? 0 : 1;
#endif
        }
    }
}

// ---------------------------------------------------------------------------
// patchtime: the same decomposition with the workgroup PERSISTENT over a run
// of K chunks (time slices).  In spmm_patchcell every (patch, chunk)
// workgroup starts from nothing: three dependent trips for the patch's cell
// list, a trip per ENTRY for its (index, weight) pairs inside the compute
// loop, the gather, a barrier, the sums -- 13.7 us of life per workgroup for
// ~1 us of work (rocprofv3 on (Time = 120, nCells): 65 % of the wave cycles
// waiting, the 14 MB patch plan re-read by every one of the 15 chunks: 200 MB
// of the 509 MB the fabric moves for 226 MB of X).  Here a workgroup reads
// its patch's metadata ONCE -- cell bases and the first 8 entries of every
// row stay in registers -- and walks its chunks with two LDS images: the X
// values of chunk c + 1 are in flight (registers) while chunk c is summed
// from LDS and stored.  One barrier per chunk, which does not drain the Y
// stores.  Same sums, same order, same bits.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier()
{
    // (not __syncthreads(): that also waits for the Y stores in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// RUNS: fields with SHORT contiguous runs behind the source axes in several
// batches -- (Time, nCells, L) with 4 <= L < 16: ten soil layers, five ice
// categories.  The same walk, a batch at a time (ceil(L / TT) chunks of one
// batch), with the results of a batch gathered in LDS and written out
// together: a lane that stored its own row's values would write 8 bytes
// every 8 L bytes (measured: 1.4-5.8 ms where this takes 0.5), the patch's
// rows x L block of Y is contiguous per tile row and leaves as whole lines.
template <typename XT, int MODE, bool FMA, int TT, int BLOCK, bool RUNS = false>
__global__ __launch_bounds__(BLOCK) void spmm_patchtime(
    const KParams p, const uint32_t flags,
    const int32_t *__restrict__ prow, const double *__restrict__ pval,
    const int32_t *__restrict__ plidx, const int32_t *__restrict__ pptr,
    const int32_t *__restrict__ ucol, const int32_t *__restrict__ row_order,
    const double *__restrict__ frac_b, const int32_t patch_rows,
    const int32_t upitch, const int64_t n_patches,
    const int64_t *__restrict__ ell_base)
{
    constexpr int NC = 2;      // cells per lane: patches hold <= 2 BLOCK cells
    constexpr int NE = 8;      // entries of a row kept in registers
    extern __shared__ __attribute__((aligned(16))) char lds[];
    double *xs = reinterpret_cast<double *>(lds);   // [2][TT][upitch]
    const int tid = threadIdx.x;
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    REMAP_CLOCK_BEGIN();
    (void)flags;
    (void)ell_base;
    const int64_t group = L / n_patches;   // runs of chunks: the slow index
    const int64_t patch = L - group * n_patches;
    const int cpw = p.rows_per_wave;       // chunks per workgroup
    // RUNS: `sub` chunks per batch, chunk c = (batch c / sub, columns
    // (c % sub) TT ...) ; else the flat column list cut every TT columns
    const uint32_t ki = p.k_inner;
    const uint32_t sub = RUNS ? (ki + TT - 1) / TT : 1;
    const uint32_t n_chunks = RUNS ? (p.K / ki) * sub : (p.K + TT - 1) / TT;
    const uint32_t c0 = static_cast<uint32_t>(group) * cpw;
    const uint32_t c1 = c0 + cpw < n_chunks ? c0 + cpw : n_chunks;
    const XT *__restrict__ X = static_cast<const XT *>(p.X);
    // RUNS with the whole batch in one chunk: ONE LDS image -- the barrier
    // in front of the batch's stores already separates a chunk's sums from
    // the next chunk's image (two images are for chunks that follow one
    // another without it) -- half the LDS, more workgroups per CU
    const bool one_image = RUNS && sub == 1;
    // RUNS: the batch's results [row][k], their mask bytes, the rows' ids
    double *out = xs + (one_image ? 1 : 2) * TT * upitch;
    int32_t *rid_lds = reinterpret_cast<int32_t *>(out + patch_rows * ki);
    uint8_t *okb = reinterpret_cast<uint8_t *>(rid_lds + patch_rows);

    // ---- once per workgroup: the patch's cells, this lane's row
    const int u0 = pptr[patch];
    const int U = pptr[patch + 1] - u0;
    int64_t cb[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        // (idle lanes name the patch's first cell -- a patch without cells
        // the plan's first: no load sits behind a branch, none leaves the
        // arrays)
        const int j = tid + q * BLOCK;
        cb[q] = cell_base(p, ucol[U > 0 ? u0 + (j < U ? j : 0) : 0]);
    }
    const int64_t slot0 = p.row_begin + patch * patch_rows;
    const int64_t local0 = patch * patch_rows;  // index into prow
    int nrows = patch_rows;
    if (slot0 + nrows > p.row_end)
        nrows = static_cast<int>(p.row_end - slot0);
    const bool has_row = tid < nrows;
    const int rr = has_row ? tid : 0;
    const int64_t i = row_order ? (int64_t)row_order[slot0 + rr] : slot0 + rr;
    const int s = prow[local0 + rr];
    const int e = has_row ? prow[local0 + rr + 1] : s;
    int32_t li[NE];
    double a[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        // clamped to an entry that exists: no load sits behind a branch
        const int jj = s + u < e ? s + u : (s < e ? s : 0);
        li[u] = plidx[jj];
        a[u] = pval[jj];
    }
    double fb = 0.0;
    if constexpr (MODE == REMAP_MODE_FRACB)
        fb = frac_b[i];
    if constexpr (RUNS) {
        if (has_row)
            rid_lds[tid] = static_cast<int32_t>(i);
    }

    // batch and first column of chunk c; column t of it exists?
    auto place = [&](uint32_t c, uint32_t &b, uint32_t &k0) {
        if constexpr (RUNS) {
            b = c / sub;
            k0 = (c - b * sub) * TT;
        } else {
            b = 0;
            k0 = c * TT;
        }
    };
    XT v[NC][TT];
    auto load = [&](uint32_t c) {
        uint32_t cbatch, k0;
        place(c, cbatch, k0);
        if constexpr (RUNS && TT % 2 == 0) {
            // a cell's run in aligned PAIRS (the host checked: even run
            // length and strides, 16-byte aligned base -- a run of 4 is two
            // 16-byte loads instead of four 8-byte ones: half the
            // instructions the texture addresser works through, the lines
            // touched twice instead of four times)
            if (p.x_pairs) {
                typedef XT pair_t __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int t = 0; t < TT; t += 2) {
                    const uint32_t k = k0 + t < ki ? k0 + t : 0u;
                    const int64_t xo =
                        static_cast<int64_t>(cbatch) * p.bsx + k;
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const pair_t pr = *reinterpret_cast<const pair_t *>(
                            X + cb[q] + xo);
                        v[q][t] = pr[0];
                        v[q][t + 1] = pr[1];
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            int64_t xo;
            if constexpr (RUNS) {
                const uint32_t k = k0 + t < ki ? k0 + t : 0u;
                xo = static_cast<int64_t>(cbatch) * p.bsx + k;
            } else {
                const uint32_t kf = k0 + t;
                const bool in = kf < p.K;
                const uint32_t b = in ? kf / ki : 0u;
                const uint32_t k = in ? kf - b * ki : 0u;
                xo = static_cast<int64_t>(b) * p.bsx + k;
            }
#pragma unroll
            for (int q = 0; q < NC; ++q)
                v[q][t] = X[cb[q] + xo];
        }
    };
    if (c0 < c1)
        load(c0);
    for (uint32_t c = c0; c < c1; ++c) {
        double *img = one_image ? xs : xs + ((c - c0) & 1) * (TT * upitch);
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int j = tid + q * BLOCK;
            if (j < U) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    img[t * upitch + j] = static_cast<double>(v[q][t]);
            }
        }
        lds_barrier();
        if (c + 1 < c1)
            load(c + 1);     // in flight while this chunk is summed
        uint32_t cbatch, k0;
        place(c, cbatch, k0);
        if (has_row) {
            double acc[TT], den[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                acc[t] = 0.0;
                den[t] = 0.0;
            }
            auto add_entry = [&](const int32_t l, const double w) {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const double x = img[t * upitch + l];
                    if constexpr (MODE == REMAP_MODE_MASKED) {
                        const bool valid = (x == x);
                        acc[t] = mul_add<FMA>(w, valid ? x : 0.0, acc[t]);
                        den[t] = den_add(w, valid ? 1.0 : 0.0, den[t]);
                    } else {
                        acc[t] = mul_add<FMA>(w, x, acc[t]);
                    }
                }
            };
#pragma unroll
            for (int u = 0; u < NE; ++u)
                if (s + u < e)
                    add_entry(li[u], a[u]);
            for (int jj = s + NE; jj < e; ++jj)      // (rows of more than 8)
                add_entry(plidx[jj], pval[jj]);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                bool ok = true;
                double y = acc[t];
                if constexpr (MODE == REMAP_MODE_FRACB) {
                    ok = fb > 0.0;
                    y = !ok ? __builtin_nan("")
                        : (fb == 1.0) ? acc[t] : acc[t] / fb;
                } else if constexpr (MODE == REMAP_MODE_MASKED) {
                    ok = den[t] > p.thr;
                    y = ok ? acc[t] / den[t] : __builtin_nan("");
                }
                if constexpr (RUNS) {
                    if (k0 + t < ki) {
                        out[tid * ki + k0 + t] = y;
                        okb[tid * ki + k0 + t] = ok ? 0 : 1;
                    }
                } else {
                    const uint32_t kf = k0 + t;
                    if (kf >= p.K)
                        continue;
                    const uint32_t b = kf / ki;
                    const uint32_t k = kf - b * ki;
                    const int64_t o =
                        i * p.ldy + static_cast<int64_t>(b) * p.bsy + k;
                    __builtin_nontemporal_store(y, p.Y + o);
#ifndef REMAP_STAMPS   // (there mask_out is the stamps buffer)
                    if (p.mask_out)
                        p.mask_out[o] = ok ? 0 : 1;
#endif
                }
            }
        }
        if constexpr (RUNS) {
            if (k0 + TT >= ki) {
                // the batch is complete: its rows x L block leaves together
                // (consecutive slots of a tile row are consecutive rows of
                // Y: whole lines).  The next write to `out` lies behind the
                // next chunk's barrier, which every lane reaches only after
                // this loop.
                lds_barrier();
                const int total = nrows * static_cast<int>(ki);
                const int64_t ob = static_cast<int64_t>(cbatch) * p.bsy;
                if (p.y_pairs) {
                    // (even run length and strides, 16-byte aligned Y: two
                    // results per store, half the store instructions)
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    for (int q = 2 * tid; q < total; q += 2 * BLOCK) {
                        const int r = q / static_cast<int>(ki);
                        const int k = q - r * static_cast<int>(ki);
                        const int64_t o =
                            static_cast<int64_t>(rid_lds[r]) * p.ldy + ob + k;
                        const d2 pr = *reinterpret_cast<const d2 *>(out + q);
                        __builtin_nontemporal_store(
                            pr, reinterpret_cast<d2 *>(p.Y + o));
#ifndef REMAP_STAMPS   // (there mask_out is the stamps buffer)
                        if (p.mask_out) {
                            p.mask_out[o] = okb[q];
                            p.mask_out[o + 1] = okb[q + 1];
                        }
#endif
                    }
                } else
                for (int q = tid; q < total; q += BLOCK) {
                    const int r = q / static_cast<int>(ki);
                    const int k = q - r * static_cast<int>(ki);
                    const int64_t o =
                        static_cast<int64_t>(rid_lds[r]) * p.ldy + ob + k;
                    __builtin_nontemporal_store(out[q], p.Y + o);
#ifndef REMAP_STAMPS   // (there mask_out is the stamps buffer)
                    if (p.mask_out)
                        p.mask_out[o] = okb[q];
#endif
                }
            }
        }
    }
    REMAP_CLOCK_END();
}
