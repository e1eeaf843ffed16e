// remap_csr.hip -- COO triplets -> CSR on the device (one-off per Remapper).
//
// Replaces `csr_matrix((S, (row - 1, col - 1)), shape=(n_b, n_a))` of
// pyremap/remapper/remap_numpy.py:134-137 (scipy: coo_tocsr + sum_duplicates):
// stable sort by (row, col), equal (row, col) entries summed left to right in
// input order, explicit zeros kept.
//
// Pipeline (all on the caller's stream, nothing synchronises):
//   pack_keys      key = row << 32 | col (0-based), count out-of-range
//   radix sort     rocPRIM radix_sort_pairs (stable) on (key, S)
//   flag_heads     1 where a new (row, col) starts
//   exclusive scan rocPRIM -> output slot of every head
//   compact_sum    one thread per head adds its run sequentially
//   build_rowptr   per row: lower_bound over the compacted rows
// rocPRIM is used as a library for the two textbook primitives; the kernels
// specific to the mapping-file semantics are written here.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "remap_common.h"

namespace remap {
namespace {

constexpr size_t kAlign = 256;

size_t align_up(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

struct Layout {
    size_t keys_in, keys_out, vals_out, head, slot, urow, temp, total;
    size_t temp_bytes;
};

int make_layout(int64_t nnz, Layout *lay)
{
    const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 1);
    size_t sort_bytes = 0, scan_bytes = 0;
    REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
        nullptr, sort_bytes, static_cast<const uint64_t *>(nullptr),
        static_cast<uint64_t *>(nullptr), static_cast<const double *>(nullptr),
        static_cast<double *>(nullptr), n, 0u, 64u)));
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        nullptr, scan_bytes, static_cast<const uint32_t *>(nullptr),
        static_cast<uint32_t *>(nullptr), 0u, n, rocprim::plus<uint32_t>())));
    lay->temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    size_t off = 0;
    lay->keys_in = off;  off += align_up(n * 8);
    lay->keys_out = off; off += align_up(n * 8);
    lay->vals_out = off; off += align_up(n * 8);
    lay->head = off;     off += align_up(n * 4);
    lay->slot = off;     off += align_up(n * 4);
    lay->urow = off;     off += align_up(n * 4);
    lay->temp = off;     off += align_up(lay->temp_bytes);
    lay->total = off;
    return REMAP_OK;
}

__global__ __launch_bounds__(kBlock) void pack_keys(
    int64_t nnz, int64_t n_rows, int64_t n_cols, int32_t base,
    const int32_t *__restrict__ row, const int32_t *__restrict__ col,
    uint64_t *__restrict__ keys, int64_t *__restrict__ bad)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz)
        return;
    const int64_t r = (int64_t)row[n] - base;
    const int64_t c = (int64_t)col[n] - base;
    if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) {
        atomicAdd(reinterpret_cast<unsigned long long *>(bad), 1ull);
        keys[n] = ~0ull;  // sorts behind every valid entry
        return;
    }
    keys[n] = (static_cast<uint64_t>(r) << 32) | static_cast<uint64_t>(c);
}

__global__ __launch_bounds__(kBlock) void flag_heads(
    int64_t nnz, const uint64_t *__restrict__ keys, uint32_t *__restrict__ head)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz)
        return;
    head[n] = (n == 0 || keys[n] != keys[n - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(kBlock) void compact_sum(
    int64_t nnz, const uint64_t *__restrict__ keys,
    const double *__restrict__ vals, const uint32_t *__restrict__ head,
    const uint32_t *__restrict__ slot, int32_t *__restrict__ col_out,
    double *__restrict__ val_out, int32_t *__restrict__ urow)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz || !head[n])
        return;
    const uint64_t key = keys[n];
    // left-to-right sum of the run of equal keys (stable sort => input order)
    double v = vals[n];
    for (int64_t m = n + 1; m < nnz && keys[m] == key; ++m)
        v = v + vals[m];
    const uint32_t o = slot[n];
    col_out[o] = static_cast<int32_t>(key & 0xffffffffull);
    val_out[o] = v;
    urow[o] = static_cast<int32_t>(key >> 32);  // -1 for rejected triplets
}

__global__ __launch_bounds__(kBlock) void build_rowptr(
    int64_t n_rows, int64_t nnz, const uint32_t *__restrict__ head,
    const uint32_t *__restrict__ slot, const int32_t *__restrict__ urow,
    int64_t *__restrict__ rowptr, int64_t *__restrict__ nnz_out)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r > n_rows)
        return;
    const int64_t n_unique =
        nnz > 0 ? (int64_t)slot[nnz - 1] + (int64_t)head[nnz - 1] : 0;
    // first compacted entry whose row is >= r; rejected triplets carry
    // row 0xffffffff (compared unsigned) and stay behind rowptr[n_rows]
    int64_t lo = 0, hi = n_unique;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (static_cast<uint32_t>(urow[mid]) < static_cast<uint64_t>(r))
            lo = mid + 1;
        else
            hi = mid;
    }
    rowptr[r] = lo;
    if (r == n_rows)
        *nnz_out = lo;
}

}  // namespace

int csr_from_coo_workspace(int64_t nnz, int64_t n_rows, size_t *bytes_out)
{
    if (!bytes_out || nnz < 0 || n_rows < 0)
        return fail(REMAP_ERR_ARG, "remap_csr_from_coo_workspace: bad args");
    Layout lay;
    const int rc = make_layout(nnz, &lay);
    if (rc != REMAP_OK)
        return rc;
    *bytes_out = lay.total;
    return REMAP_OK;
}

int csr_from_coo(int64_t n_rows, int64_t n_cols, int64_t nnz,
                 const int32_t *row, const int32_t *col, const double *S,
                 int32_t index_base, int64_t *rowptr_out, int32_t *col_out,
                 double *val_out, int64_t *nnz_out, int64_t *bad_out,
                 void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    if (n_rows < 0 || n_cols < 0 || nnz < 0)
        return fail(REMAP_ERR_ARG, "remap_csr_from_coo: negative size");
    if (n_rows >= (int64_t(1) << 31) || n_cols >= (int64_t(1) << 31) ||
        nnz >= (int64_t(1) << 32) - 1)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_csr_from_coo: sizes beyond 32-bit indices");
    if (!rowptr_out || !nnz_out || !bad_out)
        return fail(REMAP_ERR_ARG, "remap_csr_from_coo: NULL output");
    if (nnz > 0 && (!row || !col || !S || !col_out || !val_out))
        return fail(REMAP_ERR_ARG, "remap_csr_from_coo: NULL triplet array");
    Layout lay;
    int rc = make_layout(nnz, &lay);
    if (rc != REMAP_OK)
        return rc;
    if (!workspace || workspace_bytes < lay.total)
        return fail(REMAP_ERR_WORKSPACE,
                    "remap_csr_from_coo: workspace of %zu bytes, need %zu",
                    workspace_bytes, lay.total);
    char *ws = static_cast<char *>(workspace);
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws + lay.keys_in);
    uint64_t *keys_out = reinterpret_cast<uint64_t *>(ws + lay.keys_out);
    double *vals_out = reinterpret_cast<double *>(ws + lay.vals_out);
    uint32_t *head = reinterpret_cast<uint32_t *>(ws + lay.head);
    uint32_t *slot = reinterpret_cast<uint32_t *>(ws + lay.slot);
    int32_t *urow = reinterpret_cast<int32_t *>(ws + lay.urow);
    void *temp = ws + lay.temp;

    REMAP_HIP_CHECK(hipMemsetAsync(bad_out, 0, sizeof(int64_t), stream));
    const uint32_t nblk = static_cast<uint32_t>((nnz + kBlock - 1) / kBlock);
    if (nnz > 0) {
        hipLaunchKernelGGL(pack_keys, dim3(nblk), dim3(kBlock), 0, stream,
                           nnz, n_rows, n_cols, index_base, row, col, keys_in,
                           bad_out);
        REMAP_HIP_CHECK(hipGetLastError());
        size_t tb = lay.temp_bytes;
        REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
            temp, tb, static_cast<const uint64_t *>(keys_in), keys_out, S,
            vals_out, static_cast<size_t>(nnz), 0u, 64u, stream)));
        hipLaunchKernelGGL(flag_heads, dim3(nblk), dim3(kBlock), 0, stream,
                           nnz, keys_out, head);
        REMAP_HIP_CHECK(hipGetLastError());
        tb = lay.temp_bytes;
        REMAP_HIP_CHECK((rocprim::exclusive_scan(
            temp, tb, static_cast<const uint32_t *>(head), slot, 0u,
            static_cast<size_t>(nnz), rocprim::plus<uint32_t>(), stream)));
        hipLaunchKernelGGL(compact_sum, dim3(nblk), dim3(kBlock), 0, stream,
                           nnz, keys_out, vals_out, head, slot, col_out,
                           val_out, urow);
        REMAP_HIP_CHECK(hipGetLastError());
    }
    const uint32_t rblk =
        static_cast<uint32_t>((n_rows + 1 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(build_rowptr, dim3(rblk), dim3(kBlock), 0, stream,
                       n_rows, nnz, head, slot, urow, rowptr_out, nnz_out);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

}  // namespace remap

extern "C" {

int remap_csr_from_coo_workspace(int64_t nnz, int64_t n_rows,
                                 size_t *bytes_out)
{
    return remap::csr_from_coo_workspace(nnz, n_rows, bytes_out);
}

int remap_csr_from_coo(int64_t n_rows, int64_t n_cols, int64_t nnz,
                       const int32_t *row, const int32_t *col,
                       const double *S, int32_t index_base,
                       int64_t *rowptr_out, int32_t *col_out, double *val_out,
                       int64_t *nnz_out, int64_t *bad_out, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    return remap::csr_from_coo(n_rows, n_cols, nnz, row, col, S, index_base,
                               rowptr_out, col_out, val_out, nnz_out, bad_out,
                               workspace, workspace_bytes,
                               static_cast<hipStream_t>(stream));
}

}  // extern "C"
