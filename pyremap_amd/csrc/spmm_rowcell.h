// spmm_rowcell.h -- family 4: lanes ACROSS destination rows, TT fields per lane.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// rowcell: for fields whose contiguous run behind the source axes is short
// or absent -- (Time, nCells), MPAS's `timeMonthly_avg_*` 2-D output, the
// reference's most common input (tests/test_interpolate.py:57-59; flattened
// by a transpose copy at remap_numpy.py:254-256) -- and (Time, nCells, few).
// There the K fields of one source cell lie n_a elements apart, so the
// lanes-across-K kernels would read 64 different cache lines per source row.
//
// Here lane l of a workgroup owns destination row r0 + l and TT consecutive
// flat columns (time slices); the column offsets are wave-uniform (scalar
// registers), the row's (col, S) pairs are lane-private, and every X access
// is an 8-byte gather inside ONE time slice -- n_a * 8 bytes (1.9 MB for
// EC30to60): it lives in the XCD's L2 while the rows of that slice are
// computed, whatever the numbering of the source cells, because neighbouring
// destination rows run in neighbouring lanes and waves and share source
// cells.  Y stores are coalesced (consecutive rows, same column: 512 B per
// wave).  Each lane adds its row's entries one after the other in CSR order,
// multiply then add: the same bits as every other family.
//
// The K chunks (TT columns each) are the slow index of the XCD-aware work
// list, so one XCD reads a given time slice of X from HBM once.
// ---------------------------------------------------------------------------
template <typename XT, int MODE, bool FMA, int TT, int UNR>
__global__ __launch_bounds__(kBlock) void spmm_rowcell(const KParams p,
                                                       const uint32_t flags)
{
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;
    const int64_t rb = L - chunk * p.n_rowblocks;
    const int64_t r = p.row_begin + rb * kBlock + threadIdx.x;
    if (r >= p.row_end)
        return;
    const int64_t i = r;

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
    const int64_t s = p.rowptr[i];
    const int64_t e = p.rowptr[i + 1];
    double acc[TT], den[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        acc[t] = 0.0;
        den[t] = 0.0;
    }
    for (int64_t base = s; base < e; base += UNR) {
        int64_t c[UNR];
        double a[UNR];
        XT xs[UNR][TT];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            // clamped to the row's last entry: no load sits behind a branch
            const int64_t jj = base + u < e ? base + u : e - 1;
            c[u] = cell_base(p, p.col[jj]);
            a[u] = p.val[jj];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int t = 0; t < TT; ++t)
                xs[u][t] = X[c[u] + xo[t]];
        asm volatile("" ::: "memory");  // loads stay ahead of their uses
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (base + u < e) {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const double x = static_cast<double>(xs[u][t]);
                    if constexpr (MODE == REMAP_MODE_MASKED) {
                        const bool valid = (x == x);
                        acc[t] = mul_add<FMA>(a[u], valid ? x : 0.0, acc[t]);
                        den[t] = den_add(a[u], valid ? 1.0 : 0.0, den[t]);
                    } else {
                        acc[t] = mul_add<FMA>(a[u], x, acc[t]);
                    }
                }
            }
        }
    }
    double fb = 0.0;
    if constexpr (MODE == REMAP_MODE_FRACB)
        fb = p.frac_b[i];
    (void)flags;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        if (!act[t])
            continue;
        bool ok = true;
        double y = acc[t];
        if constexpr (MODE == REMAP_MODE_FRACB) {
            ok = fb > 0.0;
            y = !ok ? __builtin_nan("") : (fb == 1.0) ? acc[t] : acc[t] / fb;
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
