// spmm_rowwave.h -- family 1: wave per (row, K-chunk), vector-memory metadata.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
// ---------------------------------------------------------------------------
// rowwave: one wave per (row, K-chunk); lanes across K.  Straightforward
// version: each row costs its full dependent chain rowptr -> (col, S) -> X.
// ---------------------------------------------------------------------------
template <typename XT, int VEC, int TILES, int MODE, bool FMA, int UNROLL>
__global__ __launch_bounds__(kBlock) void spmm_rowwave(const KParams p,
                                                       const uint32_t flags)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t L = logical_block(p);
    if (L >= p.n_blocks)
        return;
    const int64_t chunk = L / p.n_rowblocks;  // chunk-major work list
    const int64_t rb = L - chunk * p.n_rowblocks;

    int64_t xoff[TILES], yoff[TILES];
    bool act[TILES];
    tile_offsets<VEC, TILES>(p, chunk, lane, xoff, yoff, act);

    const XT *__restrict__ X = static_cast<const XT *>(p.X);
    const int64_t block_row0 =
        p.row_begin + rb * (int64_t)(kWavesPerBlock * p.rows_per_wave);

    for (int r = 0; r < p.rows_per_wave; ++r) {
        // the block's waves work on adjacent rows at the same time
        const int64_t slot = block_row0 + (int64_t)r * kWavesPerBlock + wave;
        if (slot >= p.row_end)
            break;
        const int64_t i = p.row_order ? (int64_t)p.row_order[slot] : slot;
        const int64_t s = p.rowptr[i];
        const int64_t e = p.rowptr[i + 1];

        double acc[TILES][VEC];
        double den[TILES][VEC];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[t][v] = 0.0;
                den[t][v] = 0.0;
            }

        for (int64_t base = s; base < e; base += kWave) {
            const int n = (e - base) < kWave ? static_cast<int>(e - base)
                                             : kWave;
            // one coalesced load brings up to 64 (col, S) pairs of the row
            int32_t my_col = 0;
            double my_val = 0.0;
            if (lane < n) {
                my_col = p.col[base + lane];
                my_val = p.val[base + lane];
            }
            accumulate_entries<XT, VEC, TILES, MODE, FMA, UNROLL>(
                X, p.ldx, xoff, my_col, my_val, n, acc, den, p);
        }

        double fb = 0.0;
        if constexpr (MODE == REMAP_MODE_FRACB)
            fb = p.frac_b[i];
        finish_row<VEC, TILES, MODE>(p, i, fb, act, yoff, acc, den);
    }
}
