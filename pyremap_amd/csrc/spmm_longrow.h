// spmm_longrow.h -- family 9: one wave per (LONG row, TT flat columns).
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// For the pole-cap rows of a global bilinear map as ESMF makes it (362-1 442
// entries each, `engine.RemapPlan._split_long_rows`) applied to FEW fields --
// one 2-D field, a monthly (Time, lat, lon) series -- where parallelism is
// scarce.  The lanes-across-rows kernel (family 7, column-major entries)
// gives each long row to ONE lane, which walks it as one chain of dependent
// loads and adds: 1 440 rows = 23 waves on the whole chip, 15 us for a
// 362-entry row whatever the field count.
//
// Only the SUM of a row has to run in order; its loads do not.  Here a wave
// owns one row and TT columns:
//   1. stage  lanes run ACROSS the row's entries (coalesced col / val loads;
//             the X values of a cap row are one latitude circle: coalesced
//             too), multiply, and leave the products in LDS: xs[t][j];
//   2. sum    lane t adds column t's products in CSR order from LDS: the
//             bits of every other family (multiply, then add, in order).
//             The loop is the row's one unavoidable chain -- an add per
//             entry -- without a trip to memory in it.
// 1 440 rows x K / TT waves fill the chip; a 362-entry row takes ~3 us.
// LDS per wave: TT x entries x 8 bytes -- every staged value is used
// once, so for MANY fields the LDS capacity becomes the bound and the
// families that stage what neighbouring long rows SHARE take over: 11
// (spmm_longwave.h, 17-128 fields) and 7 (more) -- engine.apply_strided.
//
// `A` holds the long rows only: its row r is work slot r; row_order[r] (if
// given) names the row of Y / frac_b / mask_out it writes.
// ---------------------------------------------------------------------------
// What is staged.  Separate multiply and add (the default, scipy's bits):
// the PRODUCTS -- a * x is the very value the sum would compute first, so
// the sum phase is one LDS read and one add per entry; the masked branch
// stages a * (x or 0) and, for the normaliser, a or 0.0 (den + a ==
// fma(a, 1, den), and a term of +-0 never changes a sum that started at
// +0.0).  REMAP_FLAG_FMA: weights and values apart (the fused multiply-add
// needs both).
template <int MODE, bool FMA>
constexpr int longrow_arrays(int tt)
{
    return FMA ? tt + 1 : MODE == REMAP_MODE_MASKED ? 2 * tt : tt;
}

template <typename XT, int MODE, bool FMA, int TT>
__global__ __launch_bounds__(kWave) void spmm_longrow(const KParams p,
                                                      const uint32_t flags,
                                                      const int32_t pitch)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // [TT][pitch] products (or values); then [TT][pitch] normaliser terms
    // (masked) or [pitch] weights (FMA).  pitch = 2 (mod 32): 16-byte
    // aligned columns whose lanes hit different LDS banks
    double *xs = reinterpret_cast<double *>(lds);
    double *ex = xs + TT * pitch;
    const int lane = threadIdx.x;
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;  // chunk-major work list
    const int64_t slot = p.row_begin + (L - chunk * p.n_rowblocks);
    const int64_t s = p.rowptr[slot];
    const int n = static_cast<int>(p.rowptr[slot + 1] - s);
    (void)flags;

    // wave-uniform element offsets of this chunk's TT flat columns
    int64_t xo[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const uint32_t kf = static_cast<uint32_t>(chunk) * TT + t;
        const bool in = kf < p.K;
        const uint32_t b = in ? kf / p.k_inner : 0u;
        const uint32_t k = in ? kf - b * p.k_inner : 0u;
        xo[t] = static_cast<int64_t>(b) * p.bsx + k;
    }
    const XT *__restrict__ X = static_cast<const XT *>(p.X);

    // 1. stage: lanes across the row's entries, two rounds in flight
    for (int j0 = 0; j0 < n; j0 += 2 * kWave) {
        int jj[2];
        int64_t c[2];
        double a[2];
        XT v[2][TT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // clamped to the row's last entry: no load sits behind a branch
            jj[u] = j0 + u * kWave + lane;
            const int jc = jj[u] < n ? jj[u] : n - 1;
            c[u] = cell_base(p, p.col[s + jc]);
            a[u] = p.val[s + jc];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < TT; ++t)
                v[u][t] = X[c[u] + xo[t]];
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (jj[u] < n) {
                if constexpr (FMA)
                    ex[jj[u]] = a[u];
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const double x = static_cast<double>(v[u][t]);
                    if constexpr (FMA) {
                        xs[t * pitch + jj[u]] = x;
                    } else if constexpr (MODE == REMAP_MODE_MASKED) {
                        const bool valid = (x == x);
                        xs[t * pitch + jj[u]] = a[u] * (valid ? x : 0.0);
                        ex[t * pitch + jj[u]] = valid ? a[u] : 0.0;
                    } else {
                        xs[t * pitch + jj[u]] = a[u] * x;
                    }
                }
            }
    }
    __syncthreads();

    // 2. sum: lane t walks column t in CSR order
    if (lane >= TT)
        return;
    const uint32_t kf = static_cast<uint32_t>(chunk) * TT + lane;
    if (kf >= p.K)
        return;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double *col = xs + lane * pitch;
    const double *dcol = ex + lane * pitch;
    double acc = 0.0, den = 0.0;
    int j = 0;
    if constexpr (FMA) {
        for (; j < n; ++j) {
            const double x = col[j];
            if constexpr (MODE == REMAP_MODE_MASKED) {
                const bool valid = (x == x);
                acc = __builtin_fma(ex[j], valid ? x : 0.0, acc);
                den = den_add(ex[j], valid ? 1.0 : 0.0, den);
            } else {
                acc = __builtin_fma(ex[j], x, acc);
            }
        }
    } else {
        // eight entries' LDS reads (16 bytes each) ahead of their adds
        for (; j + 8 <= n; j += 8) {
            d2 q[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q[u] = *reinterpret_cast<const d2 *>(col + j + 2 * u);
                if constexpr (MODE == REMAP_MODE_MASKED)
                    r[u] = *reinterpret_cast<const d2 *>(dcol + j + 2 * u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = acc + q[u][0];
                acc = acc + q[u][1];
                if constexpr (MODE == REMAP_MODE_MASKED) {
                    den = den + r[u][0];
                    den = den + r[u][1];
                }
            }
        }
        for (; j < n; ++j) {
            acc = acc + col[j];
            if constexpr (MODE == REMAP_MODE_MASKED)
                den = den + dcol[j];
        }
    }

    const int64_t i = p.row_order ? static_cast<int64_t>(p.row_order[slot])
                                  : slot;
    const uint32_t b = kf / p.k_inner;
    const uint32_t k = kf - b * p.k_inner;
    bool ok = true;
    double y = acc;
    if constexpr (MODE == REMAP_MODE_FRACB) {
        const double fb = p.frac_b[i];
        ok = fb > 0.0;
        y = !ok ? __builtin_nan("") : (fb == 1.0) ? acc : acc / fb;
    } else if constexpr (MODE == REMAP_MODE_MASKED) {
        ok = den > p.thr;
        y = ok ? acc / den : __builtin_nan("");
    }
    const int64_t o = i * p.ldy + static_cast<int64_t>(b) * p.bsy + k;
    __builtin_nontemporal_store(y, p.Y + o);
#ifndef REMAP_STAMPS   // (there mask_out is the stamps buffer)
    if (p.mask_out)
        p.mask_out[o] = ok ? 0 : 1;
#endif
}
