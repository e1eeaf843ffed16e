// spmm_rowsub.h -- family 3: a sub-group of lanes per row, for K <= 32.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// rowsub: an alternative few-fields path (opt-in, tune[0] = 3: the simpler
// lane-per-(row, k) kernel measured faster on every case tried, see
// automatic_family) -- ONE 2-D field (K = 1: remap_numpy.py:240-248
// appends a unit axis), monthly (Time, nCells) fields, a handful of levels.
// There is no K to spread lanes over, so lanes go across a row's ENTRIES:
// SUB (8 or 4) consecutive lanes own one destination row and fetch its
// (col, S) pairs SUB at a time -- coalesced, where the lane-per-(row, k)
// kernel walks every row with one lane -- then gather their X values and
// form the products in parallel.
//
// The sum stays in CSR order all the same: every lane of the sub-group adds
// the sub-group's products one after the other (product j read from lane j
// by ds_bpermute), i.e. ((p0 + p1) + p2) + ... exactly as scipy's
// csr_matvecs does -- bit-identical, at SUB shuffles per SUB entries.
// REMAP_FLAG_TREE (opt-in) replaces that by lane-private partial sums and a
// butterfly (__shfl_xor) at the end of the row: a different association,
// within 1e-13 relative of the default, for callers who do not need the bits.
//
// KT flat columns are carried per pass over the row (the pairs are loaded
// once per pass), any strides: a (Time, nCells) field is addressed in place
// with k_inner = 1, batch stride n_a.
// ---------------------------------------------------------------------------
template <typename XT, int MODE, bool FMA, int SUB, bool TREE>
__global__ __launch_bounds__(kBlock) void spmm_rowsub(const KParams p,
                                                      const uint32_t flags)
{
    constexpr int KT = 4;
    constexpr int kRowsPerBlock = kBlock / SUB;
    const int lane = threadIdx.x & (kWave - 1);
    const int sl = lane & (SUB - 1);          // position in the sub-group
    const int lane0 = lane & ~(SUB - 1);      // the sub-group's first lane
    const int64_t slot = p.row_begin + (int64_t)blockIdx.x * kRowsPerBlock +
                         threadIdx.x / SUB;
    if (slot >= p.row_end || gate_closed(p))
        return;   // whole sub-groups leave together
    const int64_t i = p.row_order ? (int64_t)p.row_order[slot] : slot;
    const int64_t s = p.rowptr[i];
    const int64_t e = p.rowptr[i + 1];
    const XT *__restrict__ X = static_cast<const XT *>(p.X);
    double fb = 0.0;
    if constexpr (MODE == REMAP_MODE_FRACB)
        fb = p.frac_b[i];
    (void)flags;

    for (uint32_t kf0 = 0; kf0 < p.K; kf0 += KT) {
        int64_t xo[KT], yo[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const uint32_t kf = kf0 + t < p.K ? kf0 + t : kf0;
            const uint32_t b = kf / p.k_inner;
            const uint32_t k = kf - b * p.k_inner;
            xo[t] = (int64_t)b * p.bsx + k;
            yo[t] = (int64_t)b * p.bsy + k;
        }
        double acc[KT], den[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            acc[t] = 0.0;
            den[t] = 0.0;
        }
        for (int64_t base = s; base < e; base += SUB) {
            const int64_t jj = base + sl;
            const bool have = jj < e;
            const int32_t c = have ? p.col[jj] : 0;
            const double a = have ? p.val[jj] : 0.0;
            double pn[KT], pd[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                // (lanes without an entry read row 0: in bounds, unused)
                const double x =
                    static_cast<double>(X[(int64_t)c * p.ldx + xo[t]]);
                if constexpr (MODE == REMAP_MODE_MASKED) {
                    const bool valid = (x == x);
                    pn[t] = a * (valid ? x : 0.0);
                    pd[t] = a * (valid ? 1.0 : 0.0);
                } else {
                    pn[t] = a * x;
                    pd[t] = 0.0;
                }
            }
            if constexpr (TREE) {
                if (have) {
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        acc[t] += pn[t];
                        if constexpr (MODE == REMAP_MODE_MASKED)
                            den[t] += pd[t];
                    }
                }
            } else {
                // products in entry order, one after the other: scipy's sum
                const int n = (e - base) < SUB ? static_cast<int>(e - base)
                                               : SUB;
                for (int j = 0; j < n; ++j) {
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        acc[t] = acc[t] + __shfl(pn[t], lane0 + j, kWave);
                        if constexpr (MODE == REMAP_MODE_MASKED)
                            den[t] = den[t] + __shfl(pd[t], lane0 + j, kWave);
                    }
                }
            }
        }
        if constexpr (TREE) {
#pragma unroll
            for (int w = SUB / 2; w > 0; w >>= 1) {
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    acc[t] += __shfl_xor(acc[t], w, kWave);
                    if constexpr (MODE == REMAP_MODE_MASKED)
                        den[t] += __shfl_xor(den[t], w, kWave);
                }
            }
        }
        // every lane of the sub-group holds the sums: lane t stores column t
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            if (sl == (t & (SUB - 1)) && kf0 + t < p.K) {
                bool ok = true;
                double y = acc[t];
                if constexpr (MODE == REMAP_MODE_FRACB) {
                    ok = fb > 0.0;
                    y = ok ? acc[t] / fb : __builtin_nan("");
                } else if constexpr (MODE == REMAP_MODE_MASKED) {
                    ok = den[t] > p.thr;
                    y = ok ? acc[t] / den[t] : __builtin_nan("");
                }
                const int64_t o = i * p.ldy + yo[t];
                __builtin_nontemporal_store(y, p.Y + o);
                if (p.mask_out)
                    p.mask_out[o] = ok ? 0 : 1;
            }
        }
    }
}
