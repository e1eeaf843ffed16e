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

    // 3. the schedule this mapping gets
    plan->sched = remap_schedule();
    if (n_dims > 0 && plan->nnz > 0) {
        size_t arena_bytes = 0, ws2_bytes = 0;
        if ((rc = remap_schedule_sizes(n_b, plan->nnz, &arena_bytes,
                                       &ws2_bytes)) != REMAP_OK)
            return rc;
        void *ws2 = nullptr;
        if ((rc = own.alloc(&plan->arena, arena_bytes)) != REMAP_OK ||
            (rc = own.alloc(&ws2, ws2_bytes)) != REMAP_OK)
            return rc;
        remap_csr A;
        A.n_rows = n_b;
        A.n_cols = n_a;
        A.nnz = plan->nnz;
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
                    static_cast<void *>(plan->frac_b), plan->arena})
        if (q)
            own.release(q);
    plan->device_bytes =
        static_cast<size_t>(n_b + 1) * 8 +
        static_cast<size_t>(n_s + kCsrPad) * 12 +
        static_cast<size_t>(n_b) * 8 +
        (plan->arena ? plan->sched.arena_used : 0);
    guard.p = nullptr;
    *plan_out = plan;
    return REMAP_OK;
}

// distinct source cells a patch of the lanes-across-rows kernel may stage
// (8 fields x 8 bytes each stay under 32 KB of LDS)
constexpr int64_t kCellUmax = 512;

size_t align256(size_t n) { return (n + 255) / 256 * 256; }

int prepare_short_runs(remap_plan *plan, hipStream_t stream)
{
    if (plan->cell_arena || plan->nnz == 0 || plan->n_b == 0)
        return REMAP_OK;
    const int64_t n_b = plan->n_b, nnz = plan->nnz;
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
    int32_t ty = two_d ? 16 : 1, tx = two_d ? 16 : 256;
    int64_t h[3] = {0, 0, 0};
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
        if (h[1] <= kCellUmax || ty * tx <= 16)
            break;
        if (tx >= ty && tx > 1)
            tx /= 2;
        else
            ty /= 2;
    }
    plan->cell_arena = arena;
    own.release(arena);
    plan->cell_order =
        two_d ? reinterpret_cast<int32_t *>(base + o_order) : nullptr;
    plan->cell_ptr = reinterpret_cast<int32_t *>(base + o_ptr);
    plan->cell_ucol = reinterpret_cast<int32_t *>(base + o_ucol);
    plan->cell_rowptr = reinterpret_cast<int32_t *>(base + o_rowptr);
    plan->cell_lidx = reinterpret_cast<int32_t *>(base + o_lidx);
    plan->cell_val = reinterpret_cast<double *>(base + o_val);
    plan->cell_rows = ty * tx;
    plan->cell_umax = static_cast<int32_t>(h[1]);
    plan->cell_emax = static_cast<int32_t>(h[2]);
    plan->cell_patches = (n_b + plan->cell_rows - 1) / plan->cell_rows;
    plan->device_bytes += total;
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
                    plan->cell_arena})
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
    info_out->max_row_nnz = plan->max_row_nnz;
    info_out->family = plan->sched.family;
    info_out->group_rows = plan->sched.group_rows;
    info_out->ratio = plan->sched.ratio;
    info_out->device_bytes = plan->device_bytes;
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
    a.A.nnz = plan->nnz;
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
        a.tune[1] = f->n_batch * f->k_inner >= 16 ? 8 : 4;
        a.flags |= REMAP_FLAG_TUNE_HINT;
        return remap_apply_f64(&a, stream);
    }
    if (s.family != 0) {
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
        for (int t = 0; t < 8; ++t)
            a.tune[t] = s.tune[f->mode][t];
        a.flags |= REMAP_FLAG_TUNE_HINT;
    }
    return remap_apply_f64(&a, stream);
}

}  // extern "C"
