// spmm_rowlane.h -- family 2: lane per (row, k) for K <= 32.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// rowlane: one lane per (row, k), for K <= 32
//
// A lane walks its row UNR entries at a time: the UNR (col, S) pairs are
// fetched together (index clamped to the row's last entry: no load sits
// behind a branch), then the UNR source values together, then the sum runs
// in CSR order.  A row of <= UNR entries costs three dependent memory trips
// (row pointers -> entries -> X) instead of two per entry (K = 12 on config
// 3's map: 33.9 -> 28.7 us, K = 32: 70.5 -> 59.8; K = 1 sits on the ~10 us
// launch floor either way).
// ---------------------------------------------------------------------------
template <typename XT, int MODE, bool FMA, int UNR>
__global__ __launch_bounds__(kBlock) void spmm_rowlane(const KParams p,
                                                       const uint32_t flags)
{
    if (gate_closed(p))
        return;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t r = gid / p.K;
    const uint32_t kf = static_cast<uint32_t>(gid - r * p.K);
    if (p.row_begin + r >= p.row_end)
        return;
    const int64_t i = p.row_order ? (int64_t)p.row_order[p.row_begin + r]
                                  : p.row_begin + r;
    const uint32_t b = kf / p.k_inner;
    const uint32_t k = kf - b * p.k_inner;
    const XT *__restrict__ X =
        static_cast<const XT *>(p.X) + (int64_t)b * p.bsx + k;
    const int64_t s = p.rowptr[i];
    const int64_t e = p.rowptr[i + 1];
    double acc = 0.0, den = 0.0;
    for (int64_t base = s; base < e; base += UNR) {
        int32_t c[UNR];
        double a[UNR];
        XT xs[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t jj = base + u < e ? base + u : e - 1;
            c[u] = p.col[jj];
            a[u] = p.val[jj];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            xs[u] = X[(int64_t)c[u] * p.ldx];
        asm volatile("" ::: "memory");  // loads stay ahead of their uses
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (base + u < e) {
                const double x = static_cast<double>(xs[u]);
                if constexpr (MODE == REMAP_MODE_MASKED) {
                    const bool valid = (x == x);
                    acc = mul_add<FMA>(a[u], valid ? x : 0.0, acc);
                    den = den_add(a[u], valid ? 1.0 : 0.0, den);
                } else {
                    acc = mul_add<FMA>(a[u], x, acc);
                }
            }
        }
    }
    bool ok = true;
    double y = acc;
    if constexpr (MODE == REMAP_MODE_FRACB) {
        const double fb = p.frac_b[i];
        ok = fb > 0.0;
        y = ok ? acc / fb : __builtin_nan("");
    } else if constexpr (MODE == REMAP_MODE_MASKED) {
        ok = den > p.thr;
        y = ok ? acc / den : __builtin_nan("");
    }
    const int64_t o = i * p.ldy + (int64_t)b * p.bsy + k;
    (void)flags;
    __builtin_nontemporal_store(y, p.Y + o);
    if (p.mask_out)
        p.mask_out[o] = ok ? 0 : 1;
}
