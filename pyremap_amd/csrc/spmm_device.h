// spmm_device.h -- KParams, loads/stores, the fused epilogue, the block map.
// Part of remap_spmm.hip: included there inside namespace remap::(anonymous),
// in the order given there; not a stand-alone header.
constexpr uint32_t kBatchPerChunk = 0xffffffffu;   // KParams.bpc

struct KParams {
    const int64_t *__restrict__ rowptr;
    const int32_t *__restrict__ col;
    const double *__restrict__ val;
    const void *__restrict__ X;
    double *__restrict__ Y;
    const double *__restrict__ frac_b;
    uint8_t *__restrict__ mask_out;
    const int32_t *__restrict__ row_order;
    int64_t row_begin;
    int64_t row_end;
    int64_t ldx, bsx, ldy, bsy;
    int64_t n_rowblocks;   // row blocks per chunk
    int64_t n_blocks;      // n_rowblocks * n_chunks
    int64_t blocks_per_xcd;
    double thr;
    uint32_t K;
    uint32_t k_inner;
    uint32_t bpc;          // whole batches per K tile (0: tiles cut the flat
                           // column list every kWave * VEC columns;
                           // kBatchPerChunk: the tiles of a chunk share ONE
                           // batch, tile t holding its columns from t * CH)
    int32_t rows_per_wave;
    int32_t xcd_map;
    uint32_t x_range;      // bytes addressable from a source row base
    uint32_t y_range;      // bytes addressable from a destination row base
    int64_t src_outer;     // two non-adjacent source axes (src_fold != 0):
    uint32_t src_fold;     // cell a lives at (a / fold) * src_outer +
                           // (a % fold) * ldx; else at a * ldx
    uint32_t x_pairs;      // family 7, short level runs: a cell's run is
                           // 16-byte aligned pairs of elements (even run
                           // length and strides, aligned base): two
                           // elements per load
    uint32_t y_pairs;      // ... and the results leave two per store
    const int32_t *__restrict__ gate;  // optional device-side switch: the
    int32_t gate_value;                // launch is a no-op unless *gate ==
                                       // gate_value (remap_apply_args.gate)
#ifdef REMAP_DIAG
    int32_t diag;          // diagnostic build only (tune[6]): 1 = no Y
                           // stores, 2 = gather from the first 1024 rows
#endif
};

// Bottleneck-analysis switches exist only in the diagnostic build
// (-DREMAP_DIAG, tools/build_diag.py); the product build compiles them away.
#ifdef REMAP_DIAG
#define REMAP_DIAG_COL(p, c)                                                 \
    do {                                                                     \
        if ((p).diag & 2)                                                   \
            (c) &= 1023;                                                     \
    } while (0)
#define REMAP_DIAG_SKIP_STORE(p, y0) (((p).diag & 1) && (y0) != 1.2345e300)
// ablations of the shared form (spmm_groupshare.h; WRONG results, timing
// only): 4 = no s_barrier, 8 = no sums, 16 = no DMA sends
#define REMAP_DIAG_ON(p, bit) (((p).diag & (bit)) != 0)
#else
#define REMAP_DIAG_ON(p, bit) false
#define REMAP_DIAG_COL(p, c) do { } while (0)
#define REMAP_DIAG_SKIP_STORE(p, y0) false
#endif

#ifdef REMAP_STAMPS
// The clock the chip holds INSIDE a kernel: shader cycles (s_memtime) and
// constant-rate 100 MHz ticks (s_memrealtime) of one wave per workgroup,
// from its first to its last instruction, summed into slots 6 / 7 of the
// stamps buffer: MHz = 100 * sum(cycles) / sum(ticks) (tools/clock_state.py
// --in-kernel).  Two scalar reads at either end of a workgroup: the run time
// of this build is the product's to within a fraction of a per cent.
#define REMAP_CLOCK_BEGIN()                                                  \
    const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime();           \
    const unsigned long long ck_r0 = __builtin_amdgcn_s_memrealtime()
#define REMAP_CLOCK_END()                                                    \
    do {                                                                     \
        if (threadIdx.x == 0 && p.mask_out) {                                \
            unsigned long long *o =                                          \
                reinterpret_cast<unsigned long long *>(p.mask_out);          \
            atomicAdd(o + 6, __builtin_amdgcn_s_memtime() - ck_t0);          \
            atomicAdd(o + 7, __builtin_amdgcn_s_memrealtime() - ck_r0);      \
        }                                                                    \
    } while (0)
#else
#define REMAP_CLOCK_BEGIN() do { } while (0)
#define REMAP_CLOCK_END() do { } while (0)
#endif

template <bool FMA>
__device__ __forceinline__ double mul_add(double a, double x, double acc)
{
    if constexpr (FMA) {
        return __builtin_fma(a, x, acc);
    } else {
        // separate multiply and add (the file is built with
        // -ffp-contract=off): scipy's `y[k] += a * x[k]`
        const double prod = a * x;
        return acc + prod;
    }
}

// The masked mode's normaliser, den += a * m with m = 1.0 (valid) or 0.0
// (remap_numpy.py:265, `matrix.dot(in_mask)`): the product is EXACT (a, or a
// signed zero), so one fused multiply-add rounds exactly once, to the very
// value scipy's separate multiply and add give -- RN(a*m + den) ==
// RN(RN(a*m) + den) -- including NaN / Inf weights and the sign of a zero
// sum.  Same bits, one VALU instruction fewer, whatever REMAP_FLAG_FMA says.
__device__ __forceinline__ double den_add(double a, double m, double den)
{
    return __builtin_fma(a, m, den);
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

template <typename XT, int VEC>
struct XVec;
template <>
struct XVec<double, 1> { typedef double type; };
template <>
struct XVec<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <>
struct XVec<float, 1> { typedef float type; };
template <>
struct XVec<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };

template <typename XT, int VEC>
__device__ __forceinline__ typename XVec<XT, VEC>::type load_x(const XT *p)
{
    return *reinterpret_cast<const typename XVec<XT, VEC>::type *>(p);
}

template <typename V, int VEC>
__device__ __forceinline__ double elem(const V &v, int e)
{
    if constexpr (VEC == 1) {
        return static_cast<double>(v);
    } else {
        return static_cast<double>(v[e]);
    }
}

// Y is written once and never read back: non-temporal stores (+3..5 % on
// configs 3 and H against write-back ones, A/B in one run).  There is
// deliberately no run-time switch to plain stores here: with `if (cached)
// plain else nontemporal` on the same address LLVM merges the two stores and
// silently drops the non-temporal hint -- the first builds of this file held
// 414 plain stores and not a single `nt`.
template <int VEC>
__device__ __forceinline__ void store_y(double *p, const double (&y)[VEC])
{
#ifdef REMAP_PLAIN_STORES   // (tools/build_diag.py plain: an A/B, never shipped)
    if constexpr (VEC == 1) {
        *p = y[0];
    } else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 v;
        v[0] = y[0];
        v[1] = y[1];
        *reinterpret_cast<d2 *>(p) = v;
    }
#else
    if constexpr (VEC == 1) {
        __builtin_nontemporal_store(y[0], p);
    } else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 v;
        v[0] = y[0];
        v[1] = y[1];
        __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p));
    }
#endif
}

// ---------------------------------------------------------------------------
// pieces shared by the wave-per-row kernels
// ---------------------------------------------------------------------------

// Entries [0, n) of one row chunk are held one per lane in (my_col, my_val).
// Groups of UNROLL entries: every X load of a group is issued before the
// first use, then the group is accumulated strictly in CSR order.
template <typename XT, int VEC, int TILES, int MODE, bool FMA, int UNROLL>
__device__ __forceinline__ void accumulate_entries(
    const XT *__restrict__ X, int64_t ldx, const int64_t (&xoff)[TILES],
    int32_t my_col, double my_val, int n, double (&acc)[TILES][VEC],
    double (&den)[TILES][VEC], const KParams &p)
{
    typedef typename XVec<XT, VEC>::type xvec_t;
    for (int u0 = 0; u0 < n; u0 += UNROLL) {
        xvec_t xv[UNROLL][TILES];
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu) {
            if (u0 + uu < n) {
                int32_t c = __builtin_amdgcn_readlane(my_col, u0 + uu);
                REMAP_DIAG_COL(p, c);
                const XT *xr = X + static_cast<int64_t>(c) * ldx;
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    xv[uu][t] = load_x<XT, VEC>(xr + xoff[t]);
            }
        }
        asm volatile("" ::: "memory");  // loads stay ahead of their uses
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu) {
            if (u0 + uu < n) {
                const double a = readlane_f64(my_val, u0 + uu);
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double x = elem<xvec_t, VEC>(xv[uu][t], v);
                        if constexpr (MODE == REMAP_MODE_MASKED) {
                            const bool valid = (x == x);
                            const double xz = valid ? x : 0.0;
                            const double mz = valid ? 1.0 : 0.0;
                            acc[t][v] = mul_add<FMA>(a, xz, acc[t][v]);
                            den[t][v] = den_add(a, mz, den[t][v]);
                        } else {
                            acc[t][v] = mul_add<FMA>(a, x, acc[t][v]);
                        }
                    }
            }
        }
    }
}

// One store of a row's VEC results of tile t (and of their mask bytes).
template <int VEC>
__device__ __forceinline__ void store_row_tile(const KParams &p, int64_t o,
                                               const double (&y)[VEC],
                                               const bool (&ok)[VEC])
{
    if (REMAP_DIAG_SKIP_STORE(p, y[0]))
        return;
    store_y<VEC>(p.Y + o, y);
#ifndef REMAP_STAMPS
    if (p.mask_out) {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            p.mask_out[o + v] = ok[v] ? 0 : 1;
    }
#endif
}

// acc / fb for a WAVE-UNIFORM fb (the frac_b mode: one row, one frac_b per
// wave).  `acc / fb` as hipcc expands it is 11 VALU instructions per element
// -- v_div_scale x 2, v_rcp_f64 (quarter rate), four FMAs refining the
// reciprocal, v_mul, v_fma, v_div_fmas, v_div_fixup -- a quarter of the
// row-group kernels' whole VALU stream on entry-rich mappings (config 5: 32
// elements per wave).  For ordinary operands the two v_div_scale return their
// inputs unchanged with VCC = 0, v_div_fmas is then a plain FMA and
// v_div_fixup returns its first operand: what remains is
//     y = rcp(fb), refined twice          -- depends on fb alone: ONCE per row
//     q0 = a * y;  r = fma(-fb, q0, a);  q = fma(r, y, q0)   -- per element
// the very instructions of the full sequence on the very values, hence the
// same bits.  "Ordinary" (V_DIV_SCALE_F64's conditions for scaling nothing:
// neither operand zero or denormal, exponent(a) - exponent(fb) < 768, 1 / fb
// and a / fb no denormals, biased exponent(a) > 53; nothing for v_div_fixup
// to repair): fb in [2^-126, 2^126], |a| in [2^-800, 2^600].  fb is tested on
// the scalar side, the elements with three 32-bit VALU instructions each; one
// lane outside (a zero, an Inf, a NaN, 1e-300) and the wave divides the row
// the long way.
constexpr uint32_t kDivLoHi = 223u << 20;     // biased exponent 223 = 2^-800
constexpr uint32_t kDivSpanHi = (1624u - 223u) << 20;   // ... below 2^601

__device__ __forceinline__ bool div_fast_divisor(double fb)
{
    const uint32_t eb = (static_cast<uint32_t>(__double2hiint(fb)) >> 20) &
                        0x7ffu;
    return eb >= 1023u - 126u && eb <= 1023u + 126u;   // (fb > 0 is known)
}

__device__ __forceinline__ bool div_fast_numerator(double a)
{
    const uint32_t h = static_cast<uint32_t>(__double2hiint(a)) & 0x7fffffffu;
    return h - kDivLoHi < kDivSpanHi;
}

// The stores of one row whose index is wave-uniform: the row's base comes
// from SGPRs (one scalar multiply), a lane adds its offset with ONE 64-bit
// add per tile -- written `p.Y[i * ldy + yoff]` hipcc moved i into a VGPR and
// spent a v_mad_u64_u32, an add and a shift-add per tile and row.  `all_on`:
// every lane of every tile holds a column (the caller saw it once per wave):
// no per-row exec masks.
template <int VEC, int TILES>
__device__ __forceinline__ void store_row_uniform(
    const KParams &p, int64_t i, bool okrow, bool all_on,
    const bool (&act)[TILES], const int64_t (&yoff)[TILES],
    const double (&y)[TILES][VEC])
{
    double *const yrow = p.Y + i * p.ldy;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        if (!all_on && !act[t])
            continue;
        if (REMAP_DIAG_SKIP_STORE(p, y[t][0]))
            continue;
        store_y<VEC>(yrow + yoff[t], y[t]);
#ifndef REMAP_STAMPS
        if (p.mask_out) {
            uint8_t *const mrow = p.mask_out + i * p.ldy;
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                mrow[yoff[t] + v] = okrow ? 0 : 1;
        }
#endif
    }
}

// One row divided by a WAVE-UNIFORM number (frac_b; the per-row normaliser of
// spmm_groupmask): `okrow` says whether the row is kept (else NaN, masked).
// Three ways out, each with its own stores (merged behind one store site the
// four result registers of every way were copied twice: 8 v_mov_b64 per row
// of an epilogue of ~50 instructions).
template <int VEC, int TILES>
__device__ __forceinline__ void finish_row_uniform(
    const KParams &p, int64_t i_in, double fb_in, bool okrow_in,
    const bool (&act)[TILES], const int64_t (&yoff)[TILES],
    const double (&acc)[TILES][VEC], bool all_on = false)
{
    // fb is wave-uniform (every caller: one row per wave): scalar
    // branches, no selects -- left to itself hipcc computed the
    // division, `fb == 1.0 ? acc : quotient` and the NaN fill for every
    // lane and element and selected afterwards (15 VALU per element)
    const double fb = __hiloint2double(
        __builtin_amdgcn_readfirstlane(__double2hiint(fb_in)),
        __builtin_amdgcn_readfirstlane(__double2loint(fb_in)));
    const bool okrow = __builtin_amdgcn_readfirstlane(okrow_in ? 1 : 0) != 0;
    const int64_t i =
        (static_cast<int64_t>(__builtin_amdgcn_readfirstlane(
             static_cast<int32_t>(i_in >> 32)))
         << 32) |
        static_cast<uint32_t>(
            __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i_in)));
    // every lane computes (idle lanes of a K tail hold sums of columns
    // that exist: nothing traps)
    if (okrow && fb == 1.0) {
        // x / 1.0 == x exactly: bilinear maps skip the division
        store_row_uniform<VEC, TILES>(p, i, true, all_on, act, yoff, acc);
        return;
    }
    if (okrow) {
        bool fast = div_fast_divisor(fb);
        if (fast) {
            // one compare into an SGPR pair per element, ORed on the scalar
            // side (as `in = in && ...` hipcc nested an exec mask per
            // element, as `in & ...` it packed the flags into bytes)
            uint64_t out = 0;
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    out |= __ballot(!div_fast_numerator(acc[t][v]));
            fast = out == 0ull;
        }
        if (fast) {
            double y[TILES][VEC];
            double r = __builtin_amdgcn_rcp(fb);
            double e = __builtin_fma(-fb, r, 1.0);
            r = __builtin_fma(r, e, r);
            e = __builtin_fma(-fb, r, 1.0);
            r = __builtin_fma(r, e, r);
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double q0 = acc[t][v] * r;
                    const double rem = __builtin_fma(-fb, q0, acc[t][v]);
                    y[t][v] = __builtin_fma(rem, r, q0);
                }
            store_row_uniform<VEC, TILES>(p, i, true, all_on, act, yoff, y);
            return;
        }
    }
    // rare: a row under the mask (NaN), or operands the fast division does
    // not cover (one division after the other -- interleaved by the
    // scheduler, four 11-instruction sequences hold enough temporaries to
    // spill in the kernels that run at the register limit)
    double y[TILES][VEC];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            y[t][v] = okrow ? acc[t][v] / fb : __builtin_nan("");
            asm volatile("" : "+v"(y[t][v]));
        }
    store_row_uniform<VEC, TILES>(p, i, okrow, all_on, act, yoff, y);
}

// finish_row_uniform as it was first written -- ONE store site behind the
// three ways, the guards as nested conditions: spmm_groupmask.h, which sits at
// the register limit of three waves per SIMD, spills 8 VGPRs with the form
// above and none with this one.
template <int VEC, int TILES>
__device__ __forceinline__ void finish_row_uniform_one_site(
    const KParams &p, int64_t i, double fb_in, bool okrow_in,
    const bool (&act)[TILES], const int64_t (&yoff)[TILES],
    const double (&acc)[TILES][VEC])
{
    // fb is wave-uniform (every caller: one row per wave): scalar
    // branches, no selects -- left to itself hipcc computed the
    // division, `fb == 1.0 ? acc : quotient` and the NaN fill for every
    // lane and element and selected afterwards (15 VALU per element)
    const double fb = __hiloint2double(
        __builtin_amdgcn_readfirstlane(__double2hiint(fb_in)),
        __builtin_amdgcn_readfirstlane(__double2loint(fb_in)));
    const bool okrow = __builtin_amdgcn_readfirstlane(okrow_in ? 1 : 0) != 0;
    // every lane computes (idle lanes of a K tail hold sums of columns
    // that exist: nothing traps); ONE store site behind the three paths
    double y[TILES][VEC];
    if (!okrow) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                y[t][v] = __builtin_nan("");
    } else if (fb == 1.0) {
        // x / 1.0 == x exactly: bilinear maps skip the division
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                y[t][v] = acc[t][v];
    } else {
        bool fast = div_fast_divisor(fb);
        if (fast) {
            bool in = true;
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    in = in && div_fast_numerator(acc[t][v]);
            fast = __ballot(!in) == 0ull;
        }
        if (fast) {
            double r = __builtin_amdgcn_rcp(fb);
            double e = __builtin_fma(-fb, r, 1.0);
            r = __builtin_fma(r, e, r);
            e = __builtin_fma(-fb, r, 1.0);
            r = __builtin_fma(r, e, r);
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double q0 = acc[t][v] * r;
                    const double rem = __builtin_fma(-fb, q0, acc[t][v]);
                    y[t][v] = __builtin_fma(rem, r, q0);
                }
        } else {
            // (rare: one division after the other -- interleaved by the
            // scheduler, four 11-instruction sequences hold enough
            // temporaries to spill in the kernels that run at the register
            // limit of three waves per SIMD)
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    y[t][v] = acc[t][v] / fb;
                    asm volatile("" : "+v"(y[t][v]));
                }
        }
    }
    bool ok[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        ok[v] = okrow;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        if (act[t])
            store_row_tile<VEC>(p, i * p.ldy + yoff[t], y[t], ok);
}

// One row of the masked mode whose N elements per lane share ONE normaliser
// per LANE (spmm_timeshare.h: a lane holds N time slices of its level):
// `ok = den > thr`, `y = ok ? acc / den : NaN` (remap_numpy.py:266, 277-278).
// The reciprocal and its two refinements depend on den alone -- once per lane
// instead of once per element -- under finish_row_uniform's conditions, which
// are conditions on a LANE's operands (v_div_scale scaling nothing, VCC = 0
// in that lane, v_div_fixup with nothing to repair): a lane that keeps its
// row (`ok`) needs den in [2^-126, 2^126] and every |acc| in [2^-800,
// 2^600]; a lane that does not is filled with NaN whatever it computes.  One
// lane outside and the wave divides the row the long way.
template <int N>
__device__ __forceinline__ void finish_row_lane_den(
    const KParams &p, int64_t i, double den, const bool (&act)[N],
    const int64_t (&yoff)[N], const double (&acc)[N][1])
{
    const bool ok = den > p.thr;
    uint64_t out = __ballot(ok && !div_fast_divisor(den));
#pragma unroll
    for (int e = 0; e < N; ++e)
        out |= __ballot(ok && !div_fast_numerator(acc[e][0]));
    double y[N];
    if (out == 0ull) {
        double r = __builtin_amdgcn_rcp(den);
        double e1 = __builtin_fma(-den, r, 1.0);
        r = __builtin_fma(r, e1, r);
        e1 = __builtin_fma(-den, r, 1.0);
        r = __builtin_fma(r, e1, r);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const double q0 = acc[e][0] * r;
            const double rem = __builtin_fma(-den, q0, acc[e][0]);
            const double q = __builtin_fma(rem, r, q0);
            y[e] = ok ? q : __builtin_nan("");
        }
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            y[e] = ok ? acc[e][0] / den : __builtin_nan("");
            asm volatile("" : "+v"(y[e]));
        }
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
        if (!act[e])
            continue;
        const double y1[1] = {y[e]};
        const bool ok1[1] = {ok};
        store_row_tile<1>(p, i * p.ldy + yoff[e], y1, ok1);
    }
}

// Fused epilogue of one row: normalise, mask, store (remap_numpy.py:266-278).
template <int VEC, int TILES, int MODE>
__device__ __forceinline__ void finish_row(
    const KParams &p, int64_t i, double fb_in, const bool (&act)[TILES],
    const int64_t (&yoff)[TILES], const double (&acc)[TILES][VEC],
    const double (&den)[TILES][VEC])
{
    if constexpr (MODE == REMAP_MODE_FRACB) {
        finish_row_uniform<VEC, TILES>(p, i, fb_in, fb_in > 0.0, act, yoff,
                                       acc);
    } else {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            if (!act[t])
                continue;
            double y[VEC];
            bool ok[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if constexpr (MODE == REMAP_MODE_RAW) {
                    ok[v] = true;
                    y[v] = acc[t][v];
                } else {
                    ok[v] = den[t][v] > p.thr;
                    y[v] = ok[v] ? acc[t][v] / den[t][v] : __builtin_nan("");
                }
            }
            store_row_tile<VEC>(p, i * p.ldy + yoff[t], y, ok);
        }
    }
}

// Per-lane element offsets of a wave's K tiles (flat column -> batch, k).
template <int VEC, int TILES>
__device__ __forceinline__ void tile_offsets(
    const KParams &p, int64_t chunk, int lane, int64_t (&xoff)[TILES],
    int64_t (&yoff)[TILES], bool (&act)[TILES])
{
    constexpr int CH = kWave * VEC;
    if (p.bpc == kBatchPerChunk) {
        // one batch per chunk: level runs longer than a tile (65, 81, 101
        // levels at one element per lane) stay inside one wave
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const uint32_t k = t * CH + lane * VEC;
            act[t] = k < p.k_inner;
            xoff[t] = act[t] ? chunk * p.bsx + k : 0;
            yoff[t] = act[t] ? chunk * p.bsy + k : 0;
        }
        return;
    }
    if (p.bpc) {
        // Batch-aligned tiles: a tile holds `bpc` WHOLE batches (level
        // columns of (Time, nCells, 60 levels): 2 x 60 of a wave's 128
        // columns).  Cutting the flat column list every 128 columns instead
        // splits a batch's 480-byte run between two waves -- two XCDs, two
        // moments -- so the cache line at the cut is fetched twice and
        // written in two partial pieces.
        const uint32_t n_batch = p.K / p.k_inner;
        const uint32_t cc = lane * VEC;
        const uint32_t bi = cc / p.k_inner;
        const uint32_t k = cc - bi * p.k_inner;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const uint32_t b =
                (static_cast<uint32_t>(chunk) * TILES + t) * p.bpc + bi;
            act[t] = bi < p.bpc && b < n_batch;
            xoff[t] = act[t] ? static_cast<int64_t>(b) * p.bsx + k : 0;
            yoff[t] = act[t] ? static_cast<int64_t>(b) * p.bsy + k : 0;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const uint32_t kf = (static_cast<uint32_t>(chunk) * TILES + t) * CH +
                            lane * VEC;
        act[t] = kf < p.K;
        const uint32_t b = act[t] ? kf / p.k_inner : 0u;
        const uint32_t k = act[t] ? kf - b * p.k_inner : 0u;
        // idle lanes (K tail) read offset 0 of the row: harmless, never used
        xoff[t] = static_cast<int64_t>(b) * p.bsx + k;
        yoff[t] = static_cast<int64_t>(b) * p.bsy + k;
    }
}

// physical block -> logical block.  Blocks are dealt round-robin over the 8
// XCDs, so bid % 8 labels the XCD; give each label a contiguous range.
// A gated launch whose gate is closed does nothing: every wave reads the
// gate word (one scalar load) and leaves.
__device__ __forceinline__ bool gate_closed(const KParams &p)
{
    return p.gate != nullptr && *p.gate != p.gate_value;
}

__device__ __forceinline__ int64_t logical_block(const KParams &p)
{
    if (gate_closed(p))
        return p.n_blocks;   // callers return on L >= n_blocks
    int64_t L = blockIdx.x;
    if (p.xcd_map) {
        const int64_t xcd = L & (kXcds - 1);
        const int64_t slot = L >> 3;
        L = xcd * p.blocks_per_xcd + slot;
    }
    return L;
}

// element offset of source cell a (remap_apply_args.x_src_fold)
__device__ __forceinline__ int64_t cell_base(const KParams &p, int32_t a)
{
    if (p.src_fold == 0)
        return static_cast<int64_t>(a) * p.ldx;
    const uint32_t y = static_cast<uint32_t>(a) / p.src_fold;
    const uint32_t x = static_cast<uint32_t>(a) - y * p.src_fold;
    return static_cast<int64_t>(y) * p.src_outer +
           static_cast<int64_t>(x) * p.ldx;
}

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(v), src_lane);
    const int hi =
        __builtin_amdgcn_readlane(static_cast<int>(v >> 32), src_lane);
    return (static_cast<int64_t>(hi) << 32) |
           static_cast<int64_t>(static_cast<uint32_t>(lo));
}
