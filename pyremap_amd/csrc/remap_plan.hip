// remap_plan.hip -- the whole path behind one opaque handle.
//
// remap_plan_create  = `_load_mapping` of pyremap/remapper/remap_numpy.py:
//                      72-139 -- the mapping file's triplets become the
//                      device-resident CSR scipy would build (:134-137), plus
//                      the kernel schedule this mapping gets
//                      (remap_schedule_auto); the reference caches `_matrix`
//                      on the Remapper, a binder caches the handle.
// remap_plan_apply   = `_remap_numpy_array` (:223-297): one fused launch.
//
// The lower-level entry points (remap_csr_from_coo, remap_schedule_auto,
// remap_apply_f64) never allocate and leave every buffer to the caller --
// what a host layer with its own allocator wants (pyremap_amd/engine.py keeps
// everything in torch tensors).  This file is the other option: the library
// owns the device memory of a plan (hipMalloc / hipFree), a binder passes
// host or device arrays once and field pointers afterwards.  No compute of
// its own: it calls the entry points above.
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "remap_common.h"

struct remap_plan {
    int device = -1;
    int64_t n_a = 0, n_b = 0, nnz = 0, max_row_nnz = 0;
    int64_t *rowptr = nullptr;
    int32_t *col = nullptr;     // nnz + kCsrPad readable
    double *val = nullptr;      // nnz + kCsrPad readable
    double *frac_b = nullptr;
    void *arena = nullptr;      // the schedule's arrays
    size_t device_bytes = 0;
    remap_schedule sched;
    // destination grid (remap_plan_create) and, once
    // remap_plan_prepare_short_runs has run, the patch plan of the
    // lanes-across-rows kernel that serves (Time, nCells)-like fields
    int64_t grid_dims[2] = {0, 0};
    int32_t n_dims = 0;
    void *cell_arena = nullptr;
    int32_t *cell_order = nullptr, *cell_ptr = nullptr, *cell_ucol = nullptr,
            *cell_rowptr = nullptr, *cell_lidx = nullptr;
    double *cell_val = nullptr;
    int32_t cell_rows = 0, cell_umax = 0, cell_emax = 0;
    int64_t cell_patches = 0;
    // ... and, for mappings scheduled as row groups, the two patch plans
    // that serve (Time, nCells, 4 ... 15 levels): 256-row patches for the
    // batch-at-a-time kernel (4 <= L <= 6, engine.RemapPlan.run_cells) and
    // small LDS patches for family 5 (7 <= L < 16, run_patches)
    struct PatchSet {
        void *arena = nullptr;
        int32_t *order = nullptr, *ptr = nullptr, *ucol = nullptr,
                *rowptr = nullptr, *lidx = nullptr;
        double *val = nullptr;
        int32_t rows = 0, umax = 0, emax = 0;
        int64_t n = 0;
    };
    PatchSet run_cells, runs;
    // LONG ROWS APART (split_long_rows): rowptr / col / val above then hold
    // the mapping WITHOUT its long rows' entries (schedule and short-run
    // patches are built on that), the long rows live here: their CSR, the
    // rows they are in the whole mapping, and their patch plan with the
    // entries column-major (remap_apply_args.patch_ell_base)
    int64_t n_long = 0, long_nnz = 0, long_max_row = 0;
    int64_t *long_rowptr = nullptr;
    int32_t *long_col = nullptr, *long_ids = nullptr;
    double *long_val = nullptr;
    int32_t *long_ptr = nullptr, *long_ucol = nullptr,
            *long_prow = nullptr, *long_lidx = nullptr;
    double *long_pval = nullptr;
    int64_t *long_base = nullptr;
    int32_t long_rows = 0, long_umax = 0, long_emax = 0;
    int64_t long_patches = 0;
    // the long rows once more as a ROW-MAJOR patch plan of a few consecutive
    // rows per patch: kernel family 11 (spmm_longwave.h), 17 ... 128 fields
    int32_t *wave_ptr = nullptr, *wave_ucol = nullptr, *wave_prow = nullptr,
            *wave_lidx = nullptr;
    double *wave_val = nullptr;
    int32_t wave_rows = 0, wave_umax = 0, wave_emax = 0;
    int64_t wave_patches = 0;
};

namespace remap {
namespace {

constexpr int64_t kCsrPad = 8;   // remap_csr.csr_pad the kernels want

// device allocations of one create() call, freed unless kept
struct Owned {
    std::vector<void *> ptrs;
    size_t bytes = 0;
    ~Owned()
    {
        for (void *p : ptrs)
            (void)hipFree(p);
    }
    int alloc(void **out, size_t n)
    {
        *out = nullptr;
        const hipError_t err = hipMalloc(out, n > 0 ? n : 1);
        if (err != hipSuccess)
            return hip_fail(err, "hipMalloc");
        ptrs.push_back(*out);
        bytes += n;
        return REMAP_OK;
    }
    void release(void *p)   // ownership moves to the plan
    {
        for (auto &q : ptrs)
            if (q == p)
                q = nullptr;
    }
    void free_now(void *p)
    {
        for (auto &q : ptrs)
            if (q == p && p) {
                (void)hipFree(p);
                q = nullptr;
            }
    }
};

template <typename T>
int to_device(Owned &own, const T *src, int64_t n, bool on_host,
              hipStream_t stream, const T **out)
{
    if (!on_host) {
        *out = src;
        return REMAP_OK;
    }
    void *d = nullptr;
    const int rc = own.alloc(&d, static_cast<size_t>(n) * sizeof(T));
    if (rc != REMAP_OK)
        return rc;
    REMAP_HIP_CHECK(hipMemcpyAsync(d, src, static_cast<size_t>(n) * sizeof(T),
                                   hipMemcpyHostToDevice, stream));
    *out = static_cast<const T *>(d);
    return REMAP_OK;
}

// one wave per destination row: its entries from row rows[d] (or d) of the
// source CSR
// kinds[0 .. n) = 0 (remap_plan_apply_auto).  A kernel, not hipMemsetAsync:
// a 16-byte memset captured into a hipGraph replayed as garbage on ROCm 7.2
// (the 8-byte one of ABI 24 did not); this is four stores in any case.
__global__ void zero_words_kernel(int32_t *__restrict__ w, int n)
{
    if (static_cast<int>(threadIdx.x) < n)
        w[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void copy_rows_kernel(
    const int64_t *__restrict__ src_rowptr, const int32_t *__restrict__ rows,
    const int64_t n_dst, const int64_t *__restrict__ dst_rowptr,
    const int32_t *__restrict__ col, const double *__restrict__ val,
    int32_t *__restrict__ col_dst, double *__restrict__ val_dst)
{
    const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d >= n_dst)
        return;
    const int lane = threadIdx.x & 63;
    const int64_t s = src_rowptr[rows ? rows[d] : d];
    const int64_t o = dst_rowptr[d];
    const int64_t n = dst_rowptr[d + 1] - o;
    for (int64_t j = lane; j < n; j += 64) {
        col_dst[o + j] = col[s + j];
        val_dst[o + j] = val[s + j];
    }
}

// one wave per slot of the long rows' patch plan: its entries to their
// column-major places (entry j of slot r of patch p at base[p] + j * rows + r)
__global__ __launch_bounds__(256) void column_major_kernel(
    const int32_t *__restrict__ prow, const int32_t *__restrict__ lidx,
    const double *__restrict__ val, const int64_t *__restrict__ base,
    const int32_t rows, const int64_t n_slots, int32_t *__restrict__ lidx_t,
    double *__restrict__ val_t)
{
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_slots)
        return;
    const int lane = threadIdx.x & 63;
    const int64_t patch = g / rows;
    const int64_t r = g - patch * rows;
    const int32_t s = prow[g];
    const int32_t n = prow[g + 1] - s;
    const int64_t b = base[patch] + r;
    for (int32_t j = lane; j < n; j += 64) {
        lidx_t[b + (int64_t)j * rows] = lidx[s + j];
        val_t[b + (int64_t)j * rows] = val[s + j];
    }
}

// a row counts as LONG from this many entries on (engine.RemapPlan.LONG_ROW)
constexpr int64_t kLongRow = 96;
// LDS the lanes-across-rows kernel may take (kPatchLdsMax), 4 fields a lane
constexpr int64_t kLongUmax = 160 * 1024 / (4 * 8) - 2;
// long rows per workgroup of family 11 (engine.LONG_WAVE_ROWS) and the most
// fields it takes (engine.LONG_WAVE_MAX)
constexpr int64_t kWaveRows = 6;
constexpr int64_t kWaveMaxFields = 128;

// Mappings whose few long rows hold a large share of the entries (the pole
// caps of a global bilinear map as ESMF makes it): the plan keeps the mapping
// without those rows' entries and the long rows apart, applied by two
// launches writing disjoint rows -- engine.RemapPlan._split_long_rows, same
// criteria, same layout.  `h_rowptr`: the CSR's row pointers on the host.
// Leaves the plan untouched when the mapping has no such rows.
int split_long_rows(remap_plan *plan, Owned &own,
                    const std::vector<int64_t> &h_rowptr, hipStream_t stream)
{
    const int64_t n_b = plan->n_b;
    if (plan->max_row_nnz <= kLongRow)
        return REMAP_OK;
    std::vector<int32_t> ids;
    std::vector<int64_t> h_short(static_cast<size_t>(n_b + 1), 0);
    std::vector<int64_t> h_long(1, 0);
    int64_t long_max = 0;
    for (int64_t i = 0; i < n_b; ++i) {
        const int64_t len = h_rowptr[i + 1] - h_rowptr[i];
        if (len > kLongRow) {
            ids.push_back(static_cast<int32_t>(i));
            h_long.push_back(h_long.back() + len);
            h_short[i + 1] = h_short[i];
            if (len > long_max)
                long_max = len;
        } else {
            h_short[i + 1] = h_short[i] + len;
        }
    }
    const int64_t n_long = static_cast<int64_t>(ids.size());
    const int64_t long_nnz = h_long.back();
    const int64_t short_nnz = h_short[n_b];
    if (n_long == 0 || n_long > n_b / 8 || long_nnz < plan->nnz / 50)
        return REMAP_OK;
    int rc;
    void *p = nullptr;
    // the two CSRs
    int64_t *s_rowptr = nullptr;
    int32_t *s_col = nullptr;
    double *s_val = nullptr;
    if ((rc = own.alloc(&p, static_cast<size_t>(n_b + 1) * 8)) != REMAP_OK)
        return rc;
    s_rowptr = static_cast<int64_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(short_nnz + kCsrPad) * 4)) !=
        REMAP_OK)
        return rc;
    s_col = static_cast<int32_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(short_nnz + kCsrPad) * 8)) !=
        REMAP_OK)
        return rc;
    s_val = static_cast<double *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(n_long + 1) * 8)) != REMAP_OK)
        return rc;
    plan->long_rowptr = static_cast<int64_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(long_nnz + kCsrPad) * 4)) !=
        REMAP_OK)
        return rc;
    plan->long_col = static_cast<int32_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(long_nnz + kCsrPad) * 8)) !=
        REMAP_OK)
        return rc;
    plan->long_val = static_cast<double *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(n_long) * 4)) != REMAP_OK)
        return rc;
    plan->long_ids = static_cast<int32_t *>(p);
    REMAP_HIP_CHECK(hipMemcpyAsync(s_rowptr, h_short.data(),
                                   static_cast<size_t>(n_b + 1) * 8,
                                   hipMemcpyHostToDevice, stream));
    REMAP_HIP_CHECK(hipMemcpyAsync(plan->long_rowptr, h_long.data(),
                                   static_cast<size_t>(n_long + 1) * 8,
                                   hipMemcpyHostToDevice, stream));
    REMAP_HIP_CHECK(hipMemcpyAsync(plan->long_ids, ids.data(),
                                   static_cast<size_t>(n_long) * 4,
                                   hipMemcpyHostToDevice, stream));
    REMAP_HIP_CHECK(hipMemsetAsync(s_col + short_nnz, 0, kCsrPad * 4, stream));
    REMAP_HIP_CHECK(hipMemsetAsync(s_val + short_nnz, 0, kCsrPad * 8, stream));
    REMAP_HIP_CHECK(hipMemsetAsync(plan->long_col + long_nnz, 0, kCsrPad * 4,
                                   stream));
    REMAP_HIP_CHECK(hipMemsetAsync(plan->long_val + long_nnz, 0, kCsrPad * 8,
                                   stream));
    hipLaunchKernelGGL(copy_rows_kernel,
                       dim3(static_cast<uint32_t>((n_b + 3) / 4)), dim3(256),
                       0, stream, plan->rowptr, nullptr, n_b, s_rowptr,
                       plan->col, plan->val, s_col, s_val);
    hipLaunchKernelGGL(copy_rows_kernel,
                       dim3(static_cast<uint32_t>((n_long + 3) / 4)),
                       dim3(256), 0, stream, plan->rowptr, plan->long_ids,
                       n_long, plan->long_rowptr, plan->col, plan->val,
                       plan->long_col, plan->long_val);
    REMAP_HIP_CHECK(hipGetLastError());

    // the long rows' patch plan: 256 consecutive long rows per workgroup,
    // halved until the distinct source rows fit the LDS
    size_t ws_bytes = 0;
    if ((rc = remap_patches_workspace(n_long, long_nnz, &ws_bytes)) !=
        REMAP_OK)
        return rc;
    void *ws = nullptr, *stats = nullptr, *pl = nullptr, *pv = nullptr;
    if ((rc = own.alloc(&ws, ws_bytes)) != REMAP_OK ||
        (rc = own.alloc(&stats, 24)) != REMAP_OK)
        return rc;
    if ((rc = own.alloc(&p, static_cast<size_t>(n_long + 1) * 4)) != REMAP_OK)
        return rc;
    plan->long_ptr = static_cast<int32_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(long_nnz) * 4)) != REMAP_OK)
        return rc;
    plan->long_ucol = static_cast<int32_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(n_long + 1) * 4)) != REMAP_OK)
        return rc;
    plan->long_prow = static_cast<int32_t *>(p);
    if ((rc = own.alloc(&pl, static_cast<size_t>(long_nnz) * 4)) != REMAP_OK ||
        (rc = own.alloc(&pv, static_cast<size_t>(long_nnz) * 8)) != REMAP_OK)
        return rc;
    remap_csr A;
    A.n_rows = n_long;
    A.n_cols = plan->n_a;
    A.nnz = long_nnz;
    A.rowptr = plan->long_rowptr;
    A.col = plan->long_col;
    A.val = plan->long_val;
    A.max_row_nnz = long_max;
    A.csr_pad = kCsrPad;
    int32_t tx = 256;
    int64_t h[3] = {0, 0, 0};
    for (;;) {
        rc = remap_patches_build(&A, nullptr, 0, 1, tx, nullptr,
                                 plan->long_ptr, plan->long_ucol,
                                 plan->long_prow, static_cast<int32_t *>(pl),
                                 static_cast<double *>(pv),
                                 static_cast<int64_t *>(stats), ws, ws_bytes,
                                 stream);
        if (rc != REMAP_OK)
            return rc;
        REMAP_HIP_CHECK(hipMemcpyAsync(h, stats, 24, hipMemcpyDeviceToHost,
                                       stream));
        REMAP_HIP_CHECK(hipStreamSynchronize(stream));
        if (h[1] <= kLongUmax)
            break;
        if (tx == 1) {
            // no patch of long rows fits the LDS: the mapping stays whole
            for (void *q : {static_cast<void *>(s_rowptr),
                            static_cast<void *>(s_col),
                            static_cast<void *>(s_val),
                            static_cast<void *>(plan->long_rowptr),
                            static_cast<void *>(plan->long_col),
                            static_cast<void *>(plan->long_val),
                            static_cast<void *>(plan->long_ids),
                            static_cast<void *>(plan->long_ptr),
                            static_cast<void *>(plan->long_ucol),
                            static_cast<void *>(plan->long_prow), pl, pv, ws,
                            stats})
                own.free_now(q);
            plan->long_rowptr = nullptr;
            plan->long_col = plan->long_ids = plan->long_ptr =
                plan->long_ucol = plan->long_prow = nullptr;
            plan->long_val = nullptr;
            return REMAP_OK;
        }
        tx /= 2;
    }
    // the entries column-major inside every patch
    const int64_t n_patches = (n_long + tx - 1) / tx;
    std::vector<int32_t> h_prow(static_cast<size_t>(n_long + 1));
    REMAP_HIP_CHECK(hipMemcpyAsync(h_prow.data(), plan->long_prow,
                                   static_cast<size_t>(n_long + 1) * 4,
                                   hipMemcpyDeviceToHost, stream));
    REMAP_HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<int64_t> h_base(static_cast<size_t>(n_patches), 0);
    int64_t total = 0;
    for (int64_t q = 0; q < n_patches; ++q) {
        int64_t longest = 0;
        for (int64_t g = q * tx; g < n_long && g < (q + 1) * tx; ++g)
            if (h_prow[g + 1] - h_prow[g] > longest)
                longest = h_prow[g + 1] - h_prow[g];
        h_base[q] = total;
        total += longest * tx;
    }
    if ((rc = own.alloc(&p, static_cast<size_t>(n_patches) * 8)) != REMAP_OK)
        return rc;
    plan->long_base = static_cast<int64_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(total + kCsrPad) * 4)) !=
        REMAP_OK)
        return rc;
    plan->long_lidx = static_cast<int32_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(total + kCsrPad) * 8)) !=
        REMAP_OK)
        return rc;
    plan->long_pval = static_cast<double *>(p);
    REMAP_HIP_CHECK(hipMemcpyAsync(plan->long_base, h_base.data(),
                                   static_cast<size_t>(n_patches) * 8,
                                   hipMemcpyHostToDevice, stream));
    REMAP_HIP_CHECK(hipMemsetAsync(plan->long_lidx, 0,
                                   static_cast<size_t>(total + kCsrPad) * 4,
                                   stream));
    REMAP_HIP_CHECK(hipMemsetAsync(plan->long_pval, 0,
                                   static_cast<size_t>(total + kCsrPad) * 8,
                                   stream));
    hipLaunchKernelGGL(column_major_kernel,
                       dim3(static_cast<uint32_t>((n_long + 3) / 4)),
                       dim3(256), 0, stream, plan->long_prow,
                       static_cast<const int32_t *>(pl),
                       static_cast<const double *>(pv), plan->long_base, tx,
                       n_long, plan->long_lidx, plan->long_pval);
    REMAP_HIP_CHECK(hipGetLastError());
    REMAP_HIP_CHECK(hipStreamSynchronize(stream));
    own.free_now(pl);
    own.free_now(pv);
    // ... and row-major, kWaveRows consecutive long rows per patch (fewer
    // when their records would not fit the LDS beside the windows: per row
    // two windows of 8 cells x 512 bytes + 12 bytes per record, see
    // run_longwave): family 11 (engine.LONG_WAVE_ROWS has the measurements)
    {
        const int64_t per_row =
            2 * 8 * 512 + ((long_max + 15) / 16 * 16 + 16) * 12;
        int64_t wr = 160 * 1024 / per_row;
        if (wr > kWaveRows)
            wr = kWaveRows;
        if (wr >= 1) {
            void *w_ptr = nullptr, *w_ucol = nullptr, *w_prow = nullptr,
                 *w_lidx = nullptr, *w_val = nullptr;
            const int64_t n_wp = (n_long + wr - 1) / wr;
            if ((rc = own.alloc(&w_ptr, static_cast<size_t>(n_wp + 1) * 4)) !=
                    REMAP_OK ||
                (rc = own.alloc(&w_ucol,
                                static_cast<size_t>(long_nnz) * 4)) !=
                    REMAP_OK ||
                (rc = own.alloc(&w_prow,
                                static_cast<size_t>(n_long + 1) * 4)) !=
                    REMAP_OK ||
                (rc = own.alloc(&w_lidx,
                                static_cast<size_t>(long_nnz) * 4)) !=
                    REMAP_OK ||
                (rc = own.alloc(&w_val, static_cast<size_t>(long_nnz) * 8)) !=
                    REMAP_OK)
                return rc;
            rc = remap_patches_build(
                &A, nullptr, 0, 1, static_cast<int32_t>(wr), nullptr,
                static_cast<int32_t *>(w_ptr), static_cast<int32_t *>(w_ucol),
                static_cast<int32_t *>(w_prow),
                static_cast<int32_t *>(w_lidx), static_cast<double *>(w_val),
                static_cast<int64_t *>(stats), ws, ws_bytes, stream);
            if (rc != REMAP_OK)
                return rc;
            int64_t hw[3] = {0, 0, 0};
            REMAP_HIP_CHECK(hipMemcpyAsync(hw, stats, 24,
                                           hipMemcpyDeviceToHost, stream));
            REMAP_HIP_CHECK(hipStreamSynchronize(stream));
            plan->wave_ptr = static_cast<int32_t *>(w_ptr);
            plan->wave_ucol = static_cast<int32_t *>(w_ucol);
            plan->wave_prow = static_cast<int32_t *>(w_prow);
            plan->wave_lidx = static_cast<int32_t *>(w_lidx);
            plan->wave_val = static_cast<double *>(w_val);
            plan->wave_rows = static_cast<int32_t>(wr);
            plan->wave_umax = static_cast<int32_t>(hw[1]);
            plan->wave_emax = static_cast<int32_t>(hw[2]);
            plan->wave_patches = n_wp;
            plan->device_bytes += static_cast<size_t>(n_wp + 1) * 4 +
                                  static_cast<size_t>(n_long + 1) * 4 +
                                  static_cast<size_t>(long_nnz) * 16;
        }
    }
    own.free_now(ws);
    own.free_now(stats);
    // the plan's CSR becomes the mapping without the long rows' entries
    own.free_now(plan->rowptr);
    own.free_now(plan->col);
    own.free_now(plan->val);
    plan->rowptr = s_rowptr;
    plan->col = s_col;
    plan->val = s_val;
    plan->n_long = n_long;
    plan->long_nnz = long_nnz;
    plan->long_max_row = long_max;
    plan->long_rows = tx;
    plan->long_umax = static_cast<int32_t>(h[1]);
    plan->long_emax = static_cast<int32_t>(h[2]);
    plan->long_patches = n_patches;
    plan->max_row_nnz = 0;
    for (int64_t i = 0; i < n_b; ++i)
        if (h_short[i + 1] - h_short[i] > plan->max_row_nnz)
            plan->max_row_nnz = h_short[i + 1] - h_short[i];
    plan->device_bytes += static_cast<size_t>(n_long + 1) * 16 +
                          static_cast<size_t>(long_nnz + kCsrPad) * 16 +
                          static_cast<size_t>(n_long) * 4 +
                          static_cast<size_t>(total + kCsrPad) * 12 +
                          static_cast<size_t>(n_patches) * 8;
    return REMAP_OK;
}

int create(int64_t n_b, int64_t n_a, int64_t n_s, const int32_t *row,
           const int32_t *col, const double *S, int32_t index_base,
           const double *frac_b, bool on_host, const int64_t *grid_dims,
           int32_t n_dims, hipStream_t stream, remap_plan **plan_out)
{
    if (!plan_out)
        return fail(REMAP_ERR_ARG, "remap_plan_create: plan_out is NULL");
    *plan_out = nullptr;
    if (n_b <= 0 || n_a <= 0 || n_s < 0 || n_b >= 0x7fffffffLL ||
        n_a >= 0x7fffffffLL)
        return fail(REMAP_ERR_ARG,
                    "remap_plan_create: a (%lld, %lld) matrix with %lld "
                    "triplets", (long long)n_b, (long long)n_a,
                    (long long)n_s);
    if (!frac_b || (n_s > 0 && (!row || !col || !S)))
        return fail(REMAP_ERR_ARG, "remap_plan_create: NULL array");
    if (n_dims < 0 || n_dims > 2 || (n_dims > 0 && !grid_dims))
        return fail(REMAP_ERR_ARG, "remap_plan_create: n_dims = %d", n_dims);
    if (n_dims > 0) {
        int64_t cells = 1;
        for (int d = 0; d < n_dims; ++d)
            cells *= grid_dims[d];
        if (cells != n_b)
            return fail(REMAP_ERR_ARG,
                        "remap_plan_create: the destination grid holds %lld "
                        "cells, the mapping %lld rows", (long long)cells,
                        (long long)n_b);
    }
    Owned own;
    remap_plan *plan = new (std::nothrow) remap_plan();
    if (!plan)
        return fail(REMAP_ERR_ARG, "remap_plan_create: out of host memory");
    struct Guard {   // the handle itself, until handed over
        remap_plan *p;
        ~Guard() { delete p; }
    } guard{plan};
    REMAP_HIP_CHECK(hipGetDevice(&plan->device));
    plan->n_a = n_a;
    plan->n_b = n_b;
    plan->n_dims = n_dims;
    for (int d = 0; d < n_dims; ++d)
        plan->grid_dims[d] = grid_dims[d];

    // 1. inputs on the device
    const int32_t *d_row = nullptr, *d_col = nullptr;
    const double *d_S = nullptr;
    int rc;
    if ((rc = to_device(own, row, n_s, on_host, stream, &d_row)) != REMAP_OK ||
        (rc = to_device(own, col, n_s, on_host, stream, &d_col)) != REMAP_OK ||
        (rc = to_device(own, S, n_s, on_host, stream, &d_S)) != REMAP_OK)
        return rc;
    void *p = nullptr;
    if ((rc = own.alloc(&p, static_cast<size_t>(n_b) * 8)) != REMAP_OK)
        return rc;
    plan->frac_b = static_cast<double *>(p);
    REMAP_HIP_CHECK(hipMemcpyAsync(
        plan->frac_b, frac_b, static_cast<size_t>(n_b) * 8,
        on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, stream));

    // 2. COO -> CSR (remap_numpy.py:134-137)
    size_t ws_bytes = 0;
    if ((rc = remap_csr_from_coo_workspace(n_s, n_b, &ws_bytes)) != REMAP_OK)
        return rc;
    void *ws = nullptr, *counts = nullptr;
    if ((rc = own.alloc(&ws, ws_bytes)) != REMAP_OK ||
        (rc = own.alloc(&counts, 16)) != REMAP_OK ||
        (rc = own.alloc(&p, static_cast<size_t>(n_b + 1) * 8)) != REMAP_OK)
        return rc;
    plan->rowptr = static_cast<int64_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(n_s + kCsrPad) * 4)) !=
        REMAP_OK)
        return rc;
    plan->col = static_cast<int32_t *>(p);
    if ((rc = own.alloc(&p, static_cast<size_t>(n_s + kCsrPad) * 8)) !=
        REMAP_OK)
        return rc;
    plan->val = static_cast<double *>(p);
    REMAP_HIP_CHECK(hipMemsetAsync(plan->col, 0,
                                   static_cast<size_t>(n_s + kCsrPad) * 4,
                                   stream));
    REMAP_HIP_CHECK(hipMemsetAsync(plan->val, 0,
                                   static_cast<size_t>(n_s + kCsrPad) * 8,
                                   stream));
    int64_t *d_counts = static_cast<int64_t *>(counts);
    if (n_s > 0) {
        rc = remap_csr_from_coo(n_b, n_a, n_s, d_row, d_col, d_S, index_base,
                                plan->rowptr, plan->col, plan->val, d_counts,
                                d_counts + 1, ws, ws_bytes, stream);
        if (rc != REMAP_OK)
            return rc;
    } else {
        REMAP_HIP_CHECK(hipMemsetAsync(plan->rowptr, 0,
                                       static_cast<size_t>(n_b + 1) * 8,
                                       stream));
        REMAP_HIP_CHECK(hipMemsetAsync(counts, 0, 16, stream));
    }
    int64_t h_counts[2] = {0, 0};
    REMAP_HIP_CHECK(hipMemcpyAsync(h_counts, counts, 16,
                                   hipMemcpyDeviceToHost, stream));
    std::vector<int64_t> h_rowptr(static_cast<size_t>(n_b + 1));
    REMAP_HIP_CHECK(hipMemcpyAsync(h_rowptr.data(), plan->rowptr,
                                   static_cast<size_t>(n_b + 1) * 8,
                                   hipMemcpyDeviceToHost, stream));
    REMAP_HIP_CHECK(hipStreamSynchronize(stream));
    if (h_counts[1] != 0)
        return fail(REMAP_ERR_ARG,
                    "remap_plan_create: %lld mapping triplets have a row or "
                    "col index outside the (%lld, %lld) matrix",
                    (long long)h_counts[1], (long long)n_b, (long long)n_a);
    plan->nnz = h_counts[0];
    for (int64_t i = 0; i < n_b; ++i) {
        const int64_t len = h_rowptr[i + 1] - h_rowptr[i];
        if (len > plan->max_row_nnz)
            plan->max_row_nnz = len;
    }
    // (duplicates were summed: the entries behind nnz are stale -- zero the
    // readable pad the kernels fetch through)
    if (plan->nnz < n_s) {
        REMAP_HIP_CHECK(hipMemsetAsync(plan->col + plan->nnz, 0,
                                       kCsrPad * 4, stream));
        REMAP_HIP_CHECK(hipMemsetAsync(plan->val + plan->nnz, 0,
                                       kCsrPad * 8, stream));
    }
    own.free_now(ws);
    if (on_host) {
        own.free_now(const_cast<int32_t *>(d_row));
        own.free_now(const_cast<int32_t *>(d_col));
        own.free_now(const_cast<double *>(d_S));
    }

    // 3. long rows apart (pole caps of ESMF-made global bilinear maps), then
    //    the schedule the rest of the mapping gets
    if (n_dims > 0 && plan->nnz > 0 &&
        (rc = split_long_rows(plan, own, h_rowptr, stream)) != REMAP_OK)
        return rc;
    const int64_t sched_nnz = plan->nnz - plan->long_nnz;
    plan->sched = remap_schedule();
    if (n_dims > 0 && sched_nnz > 0) {
        size_t arena_bytes = 0, ws2_bytes = 0;
        if ((rc = remap_schedule_sizes(n_b, sched_nnz, &arena_bytes,
                                       &ws2_bytes)) != REMAP_OK)
            return rc;
        void *ws2 = nullptr;
        if ((rc = own.alloc(&plan->arena, arena_bytes)) != REMAP_OK ||
            (rc = own.alloc(&ws2, ws2_bytes)) != REMAP_OK)
            return rc;
        remap_csr A;
        A.n_rows = n_b;
        A.n_cols = n_a;
        A.nnz = sched_nnz;
        A.rowptr = plan->rowptr;
        A.col = plan->col;
        A.val = plan->val;
        A.max_row_nnz = plan->max_row_nnz;
        A.csr_pad = kCsrPad;
        rc = remap_schedule_auto(&A, plan->frac_b, grid_dims, n_dims, 0,
                                 plan->arena, arena_bytes, ws2, ws2_bytes,
                                 &plan->sched, stream);
        if (rc != REMAP_OK)
            return rc;
        REMAP_HIP_CHECK(hipStreamSynchronize(stream));
        own.free_now(ws2);
        if (plan->sched.family == 0) {
            own.free_now(plan->arena);
            plan->arena = nullptr;
        }
    }
    REMAP_HIP_CHECK(hipStreamSynchronize(stream));
    own.free_now(counts);
    // what stays belongs to the plan
    for (void *q : {static_cast<void *>(plan->rowptr),
                    static_cast<void *>(plan->col),
                    static_cast<void *>(plan->val),
                    static_cast<void *>(plan->frac_b), plan->arena,
                    static_cast<void *>(plan->long_rowptr),
                    static_cast<void *>(plan->long_col),
                    static_cast<void *>(plan->long_val),
                    static_cast<void *>(plan->long_ids),
                    static_cast<void *>(plan->long_ptr),
                    static_cast<void *>(plan->long_ucol),
                    static_cast<void *>(plan->long_prow),
                    static_cast<void *>(plan->long_lidx),
                    static_cast<void *>(plan->long_pval),
                    static_cast<void *>(plan->long_base),
                    static_cast<void *>(plan->wave_ptr),
                    static_cast<void *>(plan->wave_ucol),
                    static_cast<void *>(plan->wave_prow),
                    static_cast<void *>(plan->wave_lidx),
                    static_cast<void *>(plan->wave_val)})
        if (q)
            own.release(q);
    plan->device_bytes +=
        static_cast<size_t>(n_b + 1) * 8 +
        static_cast<size_t>(sched_nnz + kCsrPad) * 12 +
        static_cast<size_t>(n_b) * 8 +
        (plan->arena ? plan->sched.arena_used : 0);
    guard.p = nullptr;
    *plan_out = plan;
    return REMAP_OK;
}

// distinct source cells a patch of the lanes-across-rows kernel may stage
// (8 fields x 8 bytes each stay under 32 KB of LDS)
// (engine.RemapPlan.CELL_UMAX / CELL_TILE: two cells per lane of a
// 1 024-thread workgroup; 32 x 32 tiles on grids of >= 128 K cells)
constexpr int64_t kCellUmax = 2046;

size_t align256(size_t n) { return (n + 255) / 256 * 256; }

// one patch plan of the mapping without its long rows: tiles ty x tx of the
// destination grid, halved until fits(rows, umax, emax)
template <typename Fits>
int build_patch_set(remap_plan *plan, int32_t ty, int32_t tx, Fits fits,
                    bool dst_heavy_rule, remap_plan::PatchSet &out,
                    hipStream_t stream)
{
    const int64_t nnz = plan->nnz - plan->long_nnz;
    const int64_t n_b = plan->n_b;
    Owned own;
    size_t ws_bytes = 0;
    int rc = remap_patches_workspace(n_b, nnz, &ws_bytes);
    if (rc != REMAP_OK)
        return rc;
    const size_t o_order = 0;
    const size_t o_ptr = o_order + align256(static_cast<size_t>(n_b) * 4);
    const size_t o_ucol = o_ptr + align256(static_cast<size_t>(n_b + 1) * 4);
    const size_t o_rowptr = o_ucol + align256(static_cast<size_t>(nnz) * 4);
    const size_t o_lidx =
        o_rowptr + align256(static_cast<size_t>(n_b + 1) * 4);
    const size_t o_val = o_lidx + align256(static_cast<size_t>(nnz) * 4);
    const size_t total = o_val + align256(static_cast<size_t>(nnz) * 8);
    void *arena = nullptr, *ws = nullptr, *stats = nullptr;
    if ((rc = own.alloc(&arena, total)) != REMAP_OK ||
        (rc = own.alloc(&ws, ws_bytes)) != REMAP_OK ||
        (rc = own.alloc(&stats, 24)) != REMAP_OK)
        return rc;
    char *base = static_cast<char *>(arena);
    remap_csr A;
    A.n_rows = n_b;
    A.n_cols = plan->n_a;
    A.nnz = nnz;
    A.rowptr = plan->rowptr;
    A.col = plan->col;
    A.val = plan->val;
    A.max_row_nnz = plan->max_row_nnz;
    A.csr_pad = kCsrPad;
    const bool two_d = plan->n_dims == 2;
    int64_t h[3] = {0, 0, 0};
    bool to_256 = false;
    for (;;) {
        rc = remap_patches_build(
            &A, two_d ? plan->grid_dims : nullptr, 0, ty, tx,
            two_d ? reinterpret_cast<int32_t *>(base + o_order) : nullptr,
            reinterpret_cast<int32_t *>(base + o_ptr),
            reinterpret_cast<int32_t *>(base + o_ucol),
            reinterpret_cast<int32_t *>(base + o_rowptr),
            reinterpret_cast<int32_t *>(base + o_lidx),
            reinterpret_cast<double *>(base + o_val),
            static_cast<int64_t *>(stats), ws, ws_bytes, stream);
        if (rc != REMAP_OK)
            return rc;
        REMAP_HIP_CHECK(hipMemcpyAsync(h, stats, 24, hipMemcpyDeviceToHost,
                                       stream));
        REMAP_HIP_CHECK(hipStreamSynchronize(stream));
        // coarse -> fine (fewer than half as many staged source cells as
        // rows): 256-row patches (engine.RemapPlan.cell_patches)
        // ... where such large patches are few (a map with tens of
        // thousands of them keeps them: config 4's, (32, n) 3.3 vs 5.7 ms)
        // (decided ONCE, on the first plan that fits -- as
        // engine.RemapPlan.cell_patches decides it -- and then straight
        // down to <= 256 rows: re-deciding at every halving stopped at 512
        // rows where the halved plan no longer looked "few")
        const bool ok = fits((int64_t)ty * tx, h[1], h[2]);
        if (ok && !to_256 && dst_heavy_rule && ty * tx > 256 &&
            2 * h[1] < (int64_t)ty * tx &&
            (n_b + (int64_t)ty * tx - 1) / ((int64_t)ty * tx) < 1024)
            to_256 = true;
        if (ok && (!to_256 || ty * tx <= 256))
            break;
        if (ty * tx == 1)
            return REMAP_OK;   // nothing fits: the set stays empty
        if (tx >= ty && tx > 1)
            tx /= 2;
        else
            ty /= 2;
    }
    out.arena = arena;
    own.release(arena);
    out.order = two_d ? reinterpret_cast<int32_t *>(base + o_order) : nullptr;
    out.ptr = reinterpret_cast<int32_t *>(base + o_ptr);
    out.ucol = reinterpret_cast<int32_t *>(base + o_ucol);
    out.rowptr = reinterpret_cast<int32_t *>(base + o_rowptr);
    out.lidx = reinterpret_cast<int32_t *>(base + o_lidx);
    out.val = reinterpret_cast<double *>(base + o_val);
    out.rows = ty * tx;
    out.umax = static_cast<int32_t>(h[1]);
    out.emax = static_cast<int32_t>(h[2]);
    out.n = (n_b + out.rows - 1) / out.rows;
    plan->device_bytes += total;
    return REMAP_OK;
}

int prepare_short_runs(remap_plan *plan, hipStream_t stream)
{
    // (a split plan: the mapping without its long rows' entries)
    const int64_t nnz = plan->nnz - plan->long_nnz;
    if (plan->cell_arena || nnz == 0 || plan->n_b == 0)
        return REMAP_OK;
    const int64_t n_b = plan->n_b;
    const bool two_d = plan->n_dims == 2;
    int32_t ty = two_d ? 32 : 1, tx = two_d ? 32 : 1024;
    // small grids: smaller tiles, so that there still are a few hundred
    // workgroups
    while ((int64_t)ty * tx > 256 && n_b < (int64_t)128 * ty * tx) {
        if (tx >= ty)
            tx /= 2;
        else
            ty /= 2;
    }
    remap_plan::PatchSet cell;
    int rc = build_patch_set(
        plan, ty, tx,
        [](int64_t rows, int64_t umax, int64_t) {
            return umax <= kCellUmax || rows <= 16;
        },
        true, cell, stream);
    if (rc != REMAP_OK)
        return rc;
    if (!cell.arena)
        return REMAP_OK;
    // mappings scheduled as row groups: the patch plans of the short LEVEL
    // runs, (Time, nCells, 4 ... 15) -- engine.RemapPlan.run_cells /
    // run_patches have the measurements
    if (plan->sched.family == 10) {
        rc = build_patch_set(
            plan, two_d ? 16 : 1, two_d ? 16 : 256,
            [](int64_t rows, int64_t umax, int64_t) {
                return umax <= 510 || rows <= 16;
            },
            false, plan->run_cells, stream);
        if (rc == REMAP_OK)
            rc = build_patch_set(
                plan, two_d ? 4 : 1, two_d ? 8 : 32,
                [](int64_t rows, int64_t umax, int64_t emax) {
                    return (umax + 1) * 1024 + emax * 12 + rows * 24 + 32 <=
                               100 * 1024 ||
                           rows <= 4;
                },
                false, plan->runs, stream);
        if (rc != REMAP_OK) {
            (void)hipFree(cell.arena);
            if (plan->run_cells.arena)
                (void)hipFree(plan->run_cells.arena);
            plan->run_cells = remap_plan::PatchSet();
            plan->runs = remap_plan::PatchSet();
            return rc;
        }
    }
    plan->cell_arena = cell.arena;
    plan->cell_order = cell.order;
    plan->cell_ptr = cell.ptr;
    plan->cell_ucol = cell.ucol;
    plan->cell_rowptr = cell.rowptr;
    plan->cell_lidx = cell.lidx;
    plan->cell_val = cell.val;
    plan->cell_rows = cell.rows;
    plan->cell_umax = cell.umax;
    plan->cell_emax = cell.emax;
    plan->cell_patches = cell.n;
    return REMAP_OK;
}

}  // namespace
}  // namespace remap

extern "C" {

int remap_plan_prepare_short_runs(remap_plan *plan, void *stream)
{
    if (!plan)
        return remap::fail(REMAP_ERR_ARG,
                           "remap_plan_prepare_short_runs: NULL plan");
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != plan->device)
        return remap::fail(REMAP_ERR_ARG,
                           "remap_plan_prepare_short_runs: the plan lives on "
                           "device %d, the current device is %d",
                           plan->device, current);
    return remap::prepare_short_runs(plan, static_cast<hipStream_t>(stream));
}

int remap_plan_create(int64_t n_b, int64_t n_a, int64_t n_s,
                      const int32_t *row, const int32_t *col, const double *S,
                      int32_t index_base, const double *frac_b,
                      int32_t host_input, const int64_t *dst_grid_dims,
                      int32_t n_dims, void *stream, remap_plan **plan_out)
{
    return remap::create(n_b, n_a, n_s, row, col, S, index_base, frac_b,
                         host_input != 0, dst_grid_dims, n_dims,
                         static_cast<hipStream_t>(stream), plan_out);
}

void remap_plan_destroy(remap_plan *plan)
{
    if (!plan)
        return;
    // the memory belongs to the device the plan was created on, whatever
    // device the calling thread has current
    int current = -1;
    const bool switched = hipGetDevice(&current) == hipSuccess &&
                          current != plan->device &&
                          hipSetDevice(plan->device) == hipSuccess;
    for (void *p : {static_cast<void *>(plan->rowptr),
                    static_cast<void *>(plan->col),
                    static_cast<void *>(plan->val),
                    static_cast<void *>(plan->frac_b), plan->arena,
                    plan->cell_arena, plan->run_cells.arena,
                    plan->runs.arena,
                    static_cast<void *>(plan->long_rowptr),
                    static_cast<void *>(plan->long_col),
                    static_cast<void *>(plan->long_val),
                    static_cast<void *>(plan->long_ids),
                    static_cast<void *>(plan->long_ptr),
                    static_cast<void *>(plan->long_ucol),
                    static_cast<void *>(plan->long_prow),
                    static_cast<void *>(plan->long_lidx),
                    static_cast<void *>(plan->long_pval),
                    static_cast<void *>(plan->long_base),
                    static_cast<void *>(plan->wave_ptr),
                    static_cast<void *>(plan->wave_ucol),
                    static_cast<void *>(plan->wave_prow),
                    static_cast<void *>(plan->wave_lidx),
                    static_cast<void *>(plan->wave_val)})
        if (p)
            (void)hipFree(p);
    if (switched)
        (void)hipSetDevice(current);
    delete plan;
}

int remap_plan_query(const remap_plan *plan, remap_plan_info *info_out)
{
    if (!plan || !info_out)
        return remap::fail(REMAP_ERR_ARG, "remap_plan_query: NULL argument");
    info_out->n_a = plan->n_a;
    info_out->n_b = plan->n_b;
    info_out->nnz = plan->nnz;
    info_out->max_row_nnz = plan->long_max_row > plan->max_row_nnz
                                ? plan->long_max_row
                                : plan->max_row_nnz;
    info_out->family = plan->sched.family;
    info_out->group_rows = plan->sched.group_rows;
    info_out->ratio = plan->sched.ratio;
    info_out->device_bytes = plan->device_bytes;
    info_out->cell_patch_rows = plan->cell_rows;
    info_out->reserved = 0;
    return REMAP_OK;
}

int remap_plan_apply(const remap_plan *plan, const remap_field *f,
                     void *stream)
{
    if (!plan || !f)
        return remap::fail(REMAP_ERR_ARG, "remap_plan_apply: NULL argument");
    if (f->mode < 0 || f->mode > 2)
        return remap::fail(REMAP_ERR_ARG, "remap_plan_apply: mode %d",
                           f->mode);
    // A launch goes to the calling thread's current device; the plan's
    // arrays (and the stream, X and Y the caller passes) belong to the device
    // the plan was created on.  Refuse instead of launching against another
    // device's pointers.
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess) {
        (void)hipGetLastError();
        return remap::fail(REMAP_ERR_HIP, "remap_plan_apply: hipGetDevice");
    }
    if (current != plan->device)
        return remap::fail(
            REMAP_ERR_ARG,
            "remap_plan_apply: the plan lives on device %d but the calling "
            "thread's current device is %d; hipSetDevice(%d) first",
            plan->device, current, plan->device);
    remap_apply_args a = remap_apply_args();
    a.A.n_rows = plan->n_b;
    a.A.n_cols = plan->n_a;
    a.A.nnz = plan->nnz - plan->long_nnz;
    a.A.rowptr = plan->rowptr;
    a.A.col = plan->col;
    a.A.val = plan->val;
    a.A.max_row_nnz = plan->max_row_nnz;
    a.A.csr_pad = remap::kCsrPad;
    a.row_begin = 0;
    a.row_end = plan->n_b;
    a.X = f->X;
    a.x_dtype = f->x_dtype;
    a.mode = f->mode;
    a.x_row_stride = f->x_row_stride;
    a.x_batch_stride = f->x_batch_stride;
    a.Y = f->Y;
    a.y_row_stride = f->y_row_stride;
    a.y_batch_stride = f->y_batch_stride;
    a.n_batch = f->n_batch;
    a.k_inner = f->k_inner;
    a.frac_b = f->mode == REMAP_MODE_FRACB ? plan->frac_b : nullptr;
    a.threshold = f->threshold;
    a.mask_out = f->mask_out;
    a.gate = f->gate;
    a.gate_value = f->gate_value;
    a.flags = f->flags;
    const remap_schedule &s = plan->sched;
    // fields whose contiguous run behind the source axes is short, in
    // several batches -- (Time, nCells) -- take the LDS-staged
    // lanes-across-rows kernel when its patch plan has been prepared
    if (plan->cell_arena && f->k_inner < 4 && f->n_batch > 1 &&
        f->n_batch * f->k_inner >= 2) {
        a.row_order = plan->cell_order;
        a.patch_ptr = plan->cell_ptr;
        a.patch_ucol = plan->cell_ucol;
        a.patch_rowptr = plan->cell_rowptr;
        a.patch_lidx = plan->cell_lidx;
        a.patch_val = plan->cell_val;
        a.patch_rows = plan->cell_rows;
        a.patch_umax = plan->cell_umax;
        a.patch_emax = plan->cell_emax;
        a.patch_row_bytes = 1024;
        a.n_patches = plan->cell_patches;
        a.tune[0] = 7;
        a.tune[1] = 4;   // (workgroup persistent over a run of chunks:
                         // spmm_patchtime; engine.apply_strided)
        a.flags |= REMAP_FLAG_TUNE_HINT;
    } else if (s.family == 10 && f->n_batch > 1 && f->k_inner >= 4 &&
               f->k_inner < 16 && f->n_batch * f->k_inner >= 64 &&
               (f->k_inner <= 6 ? plan->run_cells.arena != nullptr
                                : plan->runs.arena != nullptr)) {
        // short LEVEL runs in several batches on a row-group mapping:
        // 4 ... 6 levels a batch at a time on 256-row patches
        // (spmm_patchtime<..., RUNS>), 7 ... 15 on small LDS patches
        // (family 5) -- engine.apply_strided
        const remap_plan::PatchSet &q =
            f->k_inner <= 6 ? plan->run_cells : plan->runs;
        a.row_order = q.order;
        a.patch_ptr = q.ptr;
        a.patch_ucol = q.ucol;
        a.patch_rowptr = q.rowptr;
        a.patch_lidx = q.lidx;
        a.patch_val = q.val;
        a.patch_rows = q.rows;
        a.patch_umax = q.umax;
        a.patch_emax = q.emax;
        a.patch_row_bytes = 1024;
        a.n_patches = q.n;
        if (f->k_inner <= 6) {
            a.tune[0] = 7;
            // 4 columns per chunk; 5 or 6 levels: 6 -- the batch is ONE
            // chunk ((80, n, 6) 0.57 -> 0.47 ms; engine.apply_strided)
            a.tune[1] = f->k_inner >= 5 ? 6 : 4;
            a.tune[2] = 2;
        } else {
            a.tune[0] = 5;
        }
        a.flags |= REMAP_FLAG_TUNE_HINT;
    } else if (s.family != 0) {
        a.row_order = s.row_order;
        a.patch_ptr = s.patch_ptr;
        a.patch_ucol = s.patch_ucol;
        a.patch_rowptr = s.patch_rowptr;
        a.patch_lidx = s.patch_lidx;
        a.patch_val = s.patch_val;
        a.patch_rows = s.patch_rows;
        a.patch_umax = s.patch_umax;
        a.patch_emax = s.patch_emax;
        a.patch_row_bytes = s.patch_row_bytes;
        a.n_patches = s.n_patches;
        a.group_meta = s.group_meta;
        a.group_col = s.group_col;
        a.group_w = s.group_w;
        a.group_mask = s.group_mask;
        a.group_rid = s.group_rid;
        a.group_frac = s.group_frac;
        a.n_groups = s.n_groups;
        a.group_rows = s.group_rows;
        a.share_meta = s.share_meta;
        a.share_col = s.share_col;
        a.share_mask = s.share_mask;
        a.share_waves = s.share_waves;
        for (int t = 0; t < 8; ++t)
            a.tune[t] = s.tune[f->mode][t];
        a.flags |= REMAP_FLAG_TUNE_HINT;
    }
    const int rc = remap_apply_f64(&a, stream);
    if (rc != REMAP_OK || plan->n_long == 0)
        return rc;
    // the long rows, apart (split_long_rows): the lanes-across-rows kernel
    // on their own patch plan, entries column-major; slot -> row of the
    // whole mapping through row_order.  Few fields per lane: a long row is
    // one dependent chain (engine.apply_strided has the measurements).
    remap_apply_args b = remap_apply_args();
    b.A.n_rows = plan->n_long;
    b.A.n_cols = plan->n_a;
    b.A.nnz = plan->long_nnz;
    b.A.rowptr = plan->long_rowptr;
    b.A.col = plan->long_col;
    b.A.val = plan->long_val;
    b.A.max_row_nnz = plan->long_max_row;
    b.A.csr_pad = remap::kCsrPad;
    b.row_begin = 0;
    b.row_end = plan->n_long;
    b.X = f->X;
    b.x_dtype = f->x_dtype;
    b.mode = f->mode;
    b.x_row_stride = f->x_row_stride;
    b.x_batch_stride = f->x_batch_stride;
    b.Y = f->Y;
    b.y_row_stride = f->y_row_stride;
    b.y_batch_stride = f->y_batch_stride;
    b.n_batch = f->n_batch;
    b.k_inner = f->k_inner;
    b.frac_b = a.frac_b;
    b.threshold = f->threshold;
    b.mask_out = f->mask_out;
    b.gate = f->gate;
    b.gate_value = f->gate_value;
    b.row_order = plan->long_ids;
    b.patch_ptr = plan->long_ptr;
    b.patch_ucol = plan->long_ucol;
    b.patch_rowptr = plan->long_prow;
    b.patch_lidx = plan->long_lidx;
    b.patch_val = plan->long_pval;
    b.patch_rows = plan->long_rows;
    b.patch_umax = plan->long_umax;
    b.patch_emax = plan->long_emax;
    b.patch_row_bytes = 1024;
    b.n_patches = plan->long_patches;
    b.patch_ell_base = plan->long_base;
    // few fields: one wave per (long row, a few columns), family 9; many:
    // the lanes-across-rows kernel on the rows' shared source cells
    // (engine.apply_strided has the measurements)
    const int64_t K = static_cast<int64_t>(f->n_batch) * f->k_inner;
    if (K <= 16) {
        b.tune[0] = 9;
    } else if (K <= remap::kWaveMaxFields && plan->wave_ptr) {
        // 17 ... 128 fields: a wave per long row, the cells a few rows
        // share sliding through LDS (family 11)
        b.patch_ptr = plan->wave_ptr;
        b.patch_ucol = plan->wave_ucol;
        b.patch_rowptr = plan->wave_prow;
        b.patch_lidx = plan->wave_lidx;
        b.patch_val = plan->wave_val;
        b.patch_rows = plan->wave_rows;
        b.patch_umax = plan->wave_umax;
        b.patch_emax = plan->wave_emax;
        b.n_patches = plan->wave_patches;
        b.patch_ell_base = nullptr;
        b.tune[0] = 11;
    } else {
        b.tune[0] = 7;
        b.tune[1] = K <= 128 ? 2 : 4;
    }
    b.flags = f->flags | REMAP_FLAG_TUNE_HINT;
    return remap_apply_f64(&b, stream);
}

int remap_plan_apply_auto(const remap_plan *plan, const remap_field *f,
                          int64_t x_elems, int32_t *kinds, void *stream)
{
    if (!plan || !f || !kinds || x_elems < 0)
        return remap::fail(REMAP_ERR_ARG,
                           "remap_plan_apply_auto: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(remap::zero_words_kernel, dim3(1), dim3(64), 0, s,
                       kinds, 4);
    REMAP_HIP_CHECK(hipGetLastError());
    int rc = remap_scan_nan_layout(f->X, f->x_dtype, plan->n_a, f->n_batch,
                                   f->k_inner, f->x_row_stride,
                                   f->x_batch_stride, kinds, stream);
    if (rc != REMAP_OK)
        return rc;
    remap_field g = *f;
    const uint32_t hints = REMAP_FLAG_CELL_MASKS | REMAP_FLAG_BATCH_MASKS;
    // the masked branch (remap_numpy.py:262-266) ...
    g.mode = REMAP_MODE_MASKED;
    const bool forms =
        plan->sched.family == 10 && plan->sched.group_rows == 8;
    if (forms) {
        // ... per-row normalisers where whole cells are missing, one per
        // lane and row where the mask is the same in every batch, per
        // element otherwise: kinds[3] names the form
        g.gate = kinds + 3;
        g.gate_value = 1;
        g.flags = (f->flags & ~hints) | REMAP_FLAG_CELL_MASKS;
        if ((rc = remap_plan_apply(plan, &g, stream)) != REMAP_OK)
            return rc;
        if (f->n_batch >= 3) {
            g.gate_value = 2;
            g.flags = (f->flags & ~hints) | REMAP_FLAG_BATCH_MASKS;
            if ((rc = remap_plan_apply(plan, &g, stream)) != REMAP_OK)
                return rc;
        }
        g.gate_value = 3;
        g.flags = f->flags & ~hints;
    } else {
        g.gate = kinds;
        g.gate_value = 1;
        g.flags = f->flags & ~hints;
    }
    if ((rc = remap_plan_apply(plan, &g, stream)) != REMAP_OK)
        return rc;
    // ... and the frac_b branch (:268-274) when the field holds no NaN
    g.mode = REMAP_MODE_FRACB;
    g.gate = kinds;
    g.gate_value = 0;
    g.flags = f->flags & ~hints;
    return remap_plan_apply(plan, &g, stream);
}

}  // extern "C"
