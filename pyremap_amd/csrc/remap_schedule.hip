// remap_schedule.hip -- builds the per-mapping schedules on the device
// (one-off per Remapper / per row shard): the row-group schedule of kernel
// family 10 (first half) and the LDS patch plan of family 5 (second half).
//
// The schedule (include/remap_hip.h, remap_apply_args.group_*) lets one wave
// compute G neighbouring destination rows over the sorted union of their
// columns.  Everything it needs is derived here from the CSR alone:
//
//   order_keys     a 2-D destination grid is walked in 2 x G/2 tiles
//                  (row-major, optionally inside super_tile^2 blocks): key
//                  of every row; 1-D destinations keep their natural order
//   radix sort     rows by key -> row of every work slot (group_rid,
//                  row_order); scatter -> slot of every row
//   entry_keys     key = ((group * n_a + col) * G + member) per CSR entry
//   radix sort     (key, S): the sorted weights ARE group_w -- the present
//                  (union entry, member) pairs in (entry, member) order
//   flag + scan    a new (group, col) starts a union entry -> its index
//   fill_union     group_col, group_mask (atomicOr of the member bits)
//   group_bounds   per group: lower_bound over the sorted keys -> first
//                  union entry / first weight (group_meta)
//   slot_info      group_rid padded to whole groups, group_frac
//
// rocPRIM supplies the two textbook primitives (radix sort, scan); nothing
// synchronises and nothing is allocated: the caller provides a workspace.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>   // rocPRIM's texture iterator calls the host memset

#include <rocprim/rocprim.hpp>

#include "remap_common.h"

namespace remap {
namespace {

constexpr size_t kAlignG = 256;

size_t align_g(size_t n) { return (n + kAlignG - 1) / kAlignG * kAlignG; }

struct GroupLayout {
    size_t row_keys_in, row_keys_out, rows_in, slot_of_row, keys_in,
        keys_out, head, uidx, temp, total;
    size_t temp_bytes;
};

int group_layout(int64_t n_rows, int64_t nnz, GroupLayout *lay)
{
    const size_t nr = static_cast<size_t>(n_rows > 0 ? n_rows : 1);
    const size_t ne = static_cast<size_t>(nnz > 0 ? nnz : 1);
    size_t a = 0, b = 0, c = 0;
    REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
        nullptr, a, static_cast<const uint64_t *>(nullptr),
        static_cast<uint64_t *>(nullptr), static_cast<const int32_t *>(nullptr),
        static_cast<int32_t *>(nullptr), nr, 0u, 64u)));
    REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
        nullptr, b, static_cast<const uint64_t *>(nullptr),
        static_cast<uint64_t *>(nullptr), static_cast<const double *>(nullptr),
        static_cast<double *>(nullptr), ne, 0u, 64u)));
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        nullptr, c, static_cast<const uint32_t *>(nullptr),
        static_cast<uint32_t *>(nullptr), 0u, ne, rocprim::plus<uint32_t>())));
    size_t d = 0;   // the keys-only sort of share_build
    REMAP_HIP_CHECK((rocprim::radix_sort_keys(
        nullptr, d, static_cast<const uint64_t *>(nullptr),
        static_cast<uint64_t *>(nullptr), ne, 0u, 64u)));
    lay->temp_bytes = a > b ? a : b;
    if (c > lay->temp_bytes)
        lay->temp_bytes = c;
    if (d > lay->temp_bytes)
        lay->temp_bytes = d;
    size_t off = 0;
    lay->row_keys_in = off;  off += align_g(nr * 8);
    lay->row_keys_out = off; off += align_g(nr * 8);
    lay->rows_in = off;      off += align_g(nr * 4);
    lay->slot_of_row = off;  off += align_g(nr * 4);
    lay->keys_in = off;      off += align_g(ne * 8);
    lay->keys_out = off;     off += align_g(ne * 8);
    lay->head = off;         off += align_g(ne * 4);
    lay->uidx = off;         off += align_g(ne * 4);
    lay->temp = off;         off += align_g(lay->temp_bytes);
    lay->total = off;
    return REMAP_OK;
}

// key of a row in the tile walk of the destination grid (my x mx cells,
// row-major numbering): 2 x gx tiles, row-major inside st x st supertiles,
// supertiles row-major over the grid
// share_waves = 2 / 4 (8-row groups): the 2 x 4 group tiles are walked inside
// 4 x 4 / 4 x 8 SUPERGROUP tiles -- 2 / 4 consecutive groups -- and those
// row-major inside the supertiles (the shared form, spmm_groupshare.h)
__global__ __launch_bounds__(kBlock) void order_keys(
    int64_t n_rows, int64_t row_offset, int64_t mx, int64_t st, int32_t G,
    int32_t share_waves, uint64_t *__restrict__ keys,
    int32_t *__restrict__ rows)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_rows)
        return;
    const int64_t gx = G / 2;
    const int64_t i = row_offset + r;
    const int64_t jy = i / mx;
    const int64_t jx = i - jy * mx;
    const int64_t nsx = (mx + st - 1) / st;
    int64_t key = ((jy / st) * nsx + jx / st) * (st * st);
    if (share_waves > 0) {
        const int64_t ty = 4, tx = 2 * share_waves;
        const int64_t ly = jy % st, lx = jx % st;
        key += ((ly / ty) * (st / tx) + lx / tx) * (ty * tx) +
               (((ly % ty) / 2) * (tx / gx) + (lx % tx) / gx) * G +
               (jy % 2) * gx + jx % gx;
    } else {
        key += (((jy % st) / 2) * (st / gx) + (jx % st) / gx) * G +
               (jy % 2) * gx + jx % gx;
    }
    keys[r] = static_cast<uint64_t>(key);
    rows[r] = static_cast<int32_t>(r);
}

__global__ __launch_bounds__(kBlock) void identity_order(
    int64_t n_rows, int32_t *__restrict__ rid, int32_t *__restrict__ slot_of)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_rows)
        return;
    rid[r] = static_cast<int32_t>(r);
    slot_of[r] = static_cast<int32_t>(r);
}

__global__ __launch_bounds__(kBlock) void invert_order(
    int64_t n_rows, const int32_t *__restrict__ rid,
    int32_t *__restrict__ slot_of, int32_t *__restrict__ row_order_out)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_rows)
        return;
    slot_of[rid[s]] = static_cast<int32_t>(s);
    if (row_order_out)
        row_order_out[s] = rid[s];
}

// one thread per row: keys of its entries
__global__ __launch_bounds__(kBlock) void entry_keys(
    int64_t n_rows, int64_t n_a, int32_t G,
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ slot_of, uint64_t *__restrict__ keys)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_rows)
        return;
    const uint64_t slot = static_cast<uint64_t>(slot_of[r]);
    const uint64_t g = slot / G, member = slot % G;
    for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e)
        keys[e] = (g * static_cast<uint64_t>(n_a) +
                   static_cast<uint64_t>(col[e])) * G + member;
}

__global__ __launch_bounds__(kBlock) void flag_union_heads(
    int64_t nnz, int32_t G, const uint64_t *__restrict__ keys,
    uint32_t *__restrict__ head)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz)
        return;
    head[n] = (n == 0 || keys[n] / G != keys[n - 1] / G) ? 1u : 0u;
}

__global__ __launch_bounds__(kBlock) void fill_union(
    int64_t nnz, int64_t n_a, int32_t G, const uint64_t *__restrict__ keys,
    const uint32_t *__restrict__ head, const uint32_t *__restrict__ uidx,
    int32_t *__restrict__ gcol, int32_t *__restrict__ gmask,
    int64_t *__restrict__ n_union_out)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz)
        return;
    const uint64_t key = keys[n];
    // head[n] = 1: entry n opens union entry uidx[n]; else it belongs to the
    // one opened before it
    const uint32_t u = head[n] ? uidx[n] : uidx[n] - 1u;
    if (head[n])
        gcol[u] = static_cast<int32_t>((key / G) % static_cast<uint64_t>(n_a));
    atomicOr(gmask + u, static_cast<int32_t>(
                            1u << static_cast<unsigned>(key % G)));
    if (n == nnz - 1)
        *n_union_out = static_cast<int64_t>(uidx[n]) + head[n];
}

__global__ __launch_bounds__(kBlock) void group_bounds(
    int64_t n_groups, int64_t nnz, int64_t n_a, int32_t G,
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ head,
    const uint32_t *__restrict__ uidx, int64_t *__restrict__ meta)
{
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g > n_groups)
        return;
    const uint64_t first = static_cast<uint64_t>(g) *
                           static_cast<uint64_t>(n_a) * G;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < first)
            lo = mid + 1;
        else
            hi = mid;
    }
    // a group's first entry always opens a union entry (head = 1 there)
    const int64_t n_union =
        nnz > 0 ? (int64_t)uidx[nnz - 1] + (int64_t)head[nnz - 1] : 0;
    meta[2 * g] = lo < nnz ? (int64_t)uidx[lo] : n_union;
    meta[2 * g + 1] = lo;
}

__global__ __launch_bounds__(kBlock) void slot_info(
    int64_t n_slots, int64_t n_rows, const double *__restrict__ frac_b,
    int32_t *__restrict__ rid, double *__restrict__ gfrac)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_slots)
        return;
    // work slots past the last row (the partial last group) name a valid row
    const int32_t r = s < n_rows ? rid[s] : static_cast<int32_t>(n_rows - 1);
    if (s >= n_rows)
        rid[s] = r;
    gfrac[s] = frac_b[r];
}

uint32_t blocks_for(int64_t n)
{
    return static_cast<uint32_t>((n + kBlock - 1) / kBlock);
}

}  // namespace

int groups_workspace(int64_t n_rows, int64_t nnz, size_t *bytes_out)
{
    if (!bytes_out || n_rows < 0 || nnz < 0)
        return fail(REMAP_ERR_ARG, "remap_groups_workspace: bad args");
    GroupLayout lay;
    const int rc = group_layout(n_rows, nnz, &lay);
    if (rc != REMAP_OK)
        return rc;
    *bytes_out = lay.total;
    return REMAP_OK;
}

int groups_build(const remap_csr *A, const double *frac_b, int32_t G,
                 const int64_t *grid_dims, int64_t row_offset,
                 int32_t super_tile, int32_t share_waves,
                 int32_t *row_order_out, int64_t *meta,
                 int32_t *gcol, int32_t *gmask, double *gw, int32_t *rid,
                 double *gfrac, int64_t *n_union_out, void *workspace,
                 size_t workspace_bytes, hipStream_t stream)
{
    if (!A || !frac_b || !meta || !gcol || !gmask || !gw || !rid || !gfrac ||
        !n_union_out)
        return fail(REMAP_ERR_ARG, "remap_groups_build: NULL argument");
    if (G != 4 && G != 8 && G != 16)
        return fail(REMAP_ERR_ARG,
                    "remap_groups_build: groups hold 4, 8 or 16 rows, not %d",
                    G);
    const int64_t n_rows = A->n_rows, nnz = A->nnz, n_a = A->n_cols;
    if (n_rows <= 0 || nnz <= 0 || n_a <= 0)
        return fail(REMAP_ERR_ARG,
                    "remap_groups_build: an empty matrix has no schedule");
    if (!A->rowptr || !A->col || !A->val)
        return fail(REMAP_ERR_ARG, "remap_groups_build: NULL CSR array");
    const int64_t n_groups = (n_rows + G - 1) / G;
    // key = ((group * n_a + col) * G + member) must fit 64 bits
    if (static_cast<long double>(n_groups) * n_a * G >= 9.0e18L ||
        nnz >= (int64_t(1) << 32) - 1)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_groups_build: mapping too large for 64-bit keys");
    int64_t st = super_tile > 0 ? super_tile : (int64_t(1) << 30);
    if (share_waves != 0 &&
        (G != 8 || (share_waves != 2 && share_waves != 4)))
        return fail(REMAP_ERR_ARG,
                    "remap_groups_build: share_waves %d: supergroups are 2 "
                    "or 4 groups of 8 rows", share_waves);
    if (share_waves != 0 && grid_dims && st % (2 * share_waves) != 0)
        return fail(REMAP_ERR_ARG,
                    "remap_groups_build: super_tile %lld is not a multiple "
                    "of the supergroup tile", (long long)st);
    if (grid_dims) {
        if (grid_dims[0] <= 0 || grid_dims[1] <= 0 || row_offset < 0 ||
            row_offset + n_rows > grid_dims[0] * grid_dims[1])
            return fail(REMAP_ERR_ARG,
                        "remap_groups_build: rows [%lld, %lld) outside the "
                        "%lld x %lld grid", (long long)row_offset,
                        (long long)(row_offset + n_rows),
                        (long long)grid_dims[0], (long long)grid_dims[1]);
        if (st % (G / 2) != 0 || st % 2 != 0)
            return fail(REMAP_ERR_ARG,
                        "remap_groups_build: super_tile %lld is not a "
                        "multiple of the group tile", (long long)st);
    }
    GroupLayout lay;
    int rc = group_layout(n_rows, nnz, &lay);
    if (rc != REMAP_OK)
        return rc;
    if (!workspace || workspace_bytes < lay.total)
        return fail(REMAP_ERR_WORKSPACE,
                    "remap_groups_build: workspace of %zu bytes, need %zu",
                    workspace_bytes, lay.total);
    char *ws = static_cast<char *>(workspace);
    uint64_t *rk_in = reinterpret_cast<uint64_t *>(ws + lay.row_keys_in);
    uint64_t *rk_out = reinterpret_cast<uint64_t *>(ws + lay.row_keys_out);
    int32_t *rows_in = reinterpret_cast<int32_t *>(ws + lay.rows_in);
    int32_t *slot_of = reinterpret_cast<int32_t *>(ws + lay.slot_of_row);
    uint64_t *k_in = reinterpret_cast<uint64_t *>(ws + lay.keys_in);
    uint64_t *k_out = reinterpret_cast<uint64_t *>(ws + lay.keys_out);
    uint32_t *head = reinterpret_cast<uint32_t *>(ws + lay.head);
    uint32_t *uidx = reinterpret_cast<uint32_t *>(ws + lay.uidx);
    void *temp = ws + lay.temp;

    // 1. work-slot order
    if (grid_dims) {
        hipLaunchKernelGGL(order_keys, dim3(blocks_for(n_rows)), dim3(kBlock),
                           0, stream, n_rows, row_offset, grid_dims[1], st, G,
                           share_waves, rk_in, rows_in);
        REMAP_HIP_CHECK(hipGetLastError());
        size_t tb = lay.temp_bytes;
        REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
            temp, tb, static_cast<const uint64_t *>(rk_in), rk_out,
            static_cast<const int32_t *>(rows_in), rid,
            static_cast<size_t>(n_rows), 0u, 64u, stream)));
        hipLaunchKernelGGL(invert_order, dim3(blocks_for(n_rows)),
                           dim3(kBlock), 0, stream, n_rows, rid, slot_of,
                           row_order_out);
        REMAP_HIP_CHECK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(identity_order, dim3(blocks_for(n_rows)),
                           dim3(kBlock), 0, stream, n_rows, rid, slot_of);
        REMAP_HIP_CHECK(hipGetLastError());
    }
    // 2. entries sorted by (group, col, member); the weights come out in
    //    group_w's order
    hipLaunchKernelGGL(entry_keys, dim3(blocks_for(n_rows)), dim3(kBlock), 0,
                       stream, n_rows, n_a, G, A->rowptr, A->col, slot_of,
                       k_in);
    REMAP_HIP_CHECK(hipGetLastError());
    size_t tb = lay.temp_bytes;
    REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
        temp, tb, static_cast<const uint64_t *>(k_in), k_out, A->val, gw,
        static_cast<size_t>(nnz), 0u, 64u, stream)));
    REMAP_HIP_CHECK(hipMemsetAsync(gw + nnz, 0, 128 * sizeof(double), stream));
    // 3. union entries
    hipLaunchKernelGGL(flag_union_heads, dim3(blocks_for(nnz)), dim3(kBlock),
                       0, stream, nnz, G, k_out, head);
    REMAP_HIP_CHECK(hipGetLastError());
    tb = lay.temp_bytes;
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        temp, tb, static_cast<const uint32_t *>(head), uidx, 0u,
        static_cast<size_t>(nnz), rocprim::plus<uint32_t>(), stream)));
    REMAP_HIP_CHECK(hipMemsetAsync(gcol, 0, (nnz + 32) * sizeof(int32_t),
                                   stream));
    REMAP_HIP_CHECK(hipMemsetAsync(gmask, 0, (nnz + 32) * sizeof(int32_t),
                                   stream));
    hipLaunchKernelGGL(fill_union, dim3(blocks_for(nnz)), dim3(kBlock), 0,
                       stream, nnz, n_a, G, k_out, head, uidx, gcol, gmask,
                       n_union_out);
    REMAP_HIP_CHECK(hipGetLastError());
    // 4. per-group bounds, per-slot row ids and frac_b
    hipLaunchKernelGGL(group_bounds, dim3(blocks_for(n_groups + 1)),
                       dim3(kBlock), 0, stream, n_groups, nnz, n_a, G, k_out,
                       head, uidx, meta);
    REMAP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(slot_info, dim3(blocks_for(n_groups * G)), dim3(kBlock),
                       0, stream, n_groups * G, n_rows, frac_b, rid, gfrac);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

// ---------------------------------------------------------------------------
// The shared union lists of the shared form of family 10 (spmm_groupshare.h):
// the same derivation as groups_build's steps 2 - 4 with supergroups of SR =
// 8 * share_waves work slots of an EXISTING 8-row schedule in place of its
// groups -- slot of every row from group_rid, keys ((supergroup * n_a + col)
// * SR + member), radix sort (keys only: the weights stay the 8-row
// schedule's), head flags, scan, union entries and member masks, bounds.
// ---------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(kBlock) void slots_of_rows(
    int64_t n_rows, const int32_t *__restrict__ rid,
    int32_t *__restrict__ slot_of)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_rows)
        return;
    slot_of[rid[s]] = static_cast<int32_t>(s);
}

}  // namespace

int share_build(const remap_csr *A, const int32_t *group_rid,
                int32_t share_waves, int64_t *meta, int32_t *scol,
                int32_t *smask, int64_t *n_union_out, void *workspace,
                size_t workspace_bytes, hipStream_t stream)
{
    if (!A || !group_rid || !meta || !scol || !smask || !n_union_out)
        return fail(REMAP_ERR_ARG, "remap_share_build: NULL argument");
    if (share_waves != 2 && share_waves != 4)
        return fail(REMAP_ERR_ARG,
                    "remap_share_build: supergroups hold 2 or 4 groups, "
                    "not %d", share_waves);
    const int32_t SR = 8 * share_waves;
    const int64_t n_rows = A->n_rows, nnz = A->nnz, n_a = A->n_cols;
    if (n_rows <= 0 || nnz <= 0 || n_a <= 0)
        return fail(REMAP_ERR_ARG,
                    "remap_share_build: an empty matrix has no schedule");
    if (!A->rowptr || !A->col)
        return fail(REMAP_ERR_ARG, "remap_share_build: NULL CSR array");
    const int64_t n_super = (n_rows + SR - 1) / SR;
    if (static_cast<long double>(n_super) * n_a * SR >= 9.0e18L ||
        nnz >= (int64_t(1) << 32) - 1)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_share_build: mapping too large for 64-bit keys");
    GroupLayout lay;
    const int rc = group_layout(n_rows, nnz, &lay);
    if (rc != REMAP_OK)
        return rc;
    if (!workspace || workspace_bytes < lay.total)
        return fail(REMAP_ERR_WORKSPACE,
                    "remap_share_build: workspace of %zu bytes, need %zu",
                    workspace_bytes, lay.total);
    char *ws = static_cast<char *>(workspace);
    int32_t *slot_of = reinterpret_cast<int32_t *>(ws + lay.slot_of_row);
    uint64_t *k_in = reinterpret_cast<uint64_t *>(ws + lay.keys_in);
    uint64_t *k_out = reinterpret_cast<uint64_t *>(ws + lay.keys_out);
    uint32_t *head = reinterpret_cast<uint32_t *>(ws + lay.head);
    uint32_t *uidx = reinterpret_cast<uint32_t *>(ws + lay.uidx);
    void *temp = ws + lay.temp;

    hipLaunchKernelGGL(slots_of_rows, dim3(blocks_for(n_rows)), dim3(kBlock),
                       0, stream, n_rows, group_rid, slot_of);
    REMAP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(entry_keys, dim3(blocks_for(n_rows)), dim3(kBlock), 0,
                       stream, n_rows, n_a, SR, A->rowptr, A->col, slot_of,
                       k_in);
    REMAP_HIP_CHECK(hipGetLastError());
    size_t tb = lay.temp_bytes;
    REMAP_HIP_CHECK((rocprim::radix_sort_keys(
        temp, tb, static_cast<const uint64_t *>(k_in), k_out,
        static_cast<size_t>(nnz), 0u, 64u, stream)));
    hipLaunchKernelGGL(flag_union_heads, dim3(blocks_for(nnz)), dim3(kBlock),
                       0, stream, nnz, SR, k_out, head);
    REMAP_HIP_CHECK(hipGetLastError());
    tb = lay.temp_bytes;
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        temp, tb, static_cast<const uint32_t *>(head), uidx, 0u,
        static_cast<size_t>(nnz), rocprim::plus<uint32_t>(), stream)));
    REMAP_HIP_CHECK(hipMemsetAsync(scol, 0, (nnz + 256) * sizeof(int32_t),
                                   stream));
    REMAP_HIP_CHECK(hipMemsetAsync(smask, 0, (nnz + 256) * sizeof(int32_t),
                                   stream));
    hipLaunchKernelGGL(fill_union, dim3(blocks_for(nnz)), dim3(kBlock), 0,
                       stream, nnz, n_a, SR, k_out, head, uidx, scol, smask,
                       n_union_out);
    REMAP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(group_bounds, dim3(blocks_for(n_super + 1)),
                       dim3(kBlock), 0, stream, n_super, nnz, n_a, SR, k_out,
                       head, uidx, meta);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

}  // namespace remap

extern "C" {

int remap_groups_workspace(int64_t n_rows, int64_t nnz, size_t *bytes_out)
{
    return remap::groups_workspace(n_rows, nnz, bytes_out);
}

int remap_groups_build(const remap_csr *A, const double *frac_b,
                       int32_t group_rows, const int64_t *grid_dims,
                       int64_t row_offset, int32_t super_tile,
                       int32_t share_waves,
                       int32_t *row_order_out, int64_t *group_meta,
                       int32_t *group_col, int32_t *group_mask,
                       double *group_w, int32_t *group_rid,
                       double *group_frac, int64_t *n_union_out,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    return remap::groups_build(A, frac_b, group_rows, grid_dims, row_offset,
                               super_tile, share_waves, row_order_out,
                               group_meta,
                               group_col, group_mask, group_w, group_rid,
                               group_frac, n_union_out, workspace,
                               workspace_bytes,
                               static_cast<hipStream_t>(stream));
}

int remap_share_build(const remap_csr *A, const int32_t *group_rid,
                      int32_t share_waves, int64_t *share_meta,
                      int32_t *share_col, int32_t *share_mask,
                      int64_t *n_union_out, void *workspace,
                      size_t workspace_bytes, void *stream)
{
    return remap::share_build(A, group_rid, share_waves, share_meta,
                              share_col, share_mask, n_union_out, workspace,
                              workspace_bytes,
                              static_cast<hipStream_t>(stream));
}

}  // extern "C"

// ===========================================================================
// The LDS patch plan of kernel family 5 (remap_apply_args.patch_*), built on
// the device the same way:
//
//   tile_keys      rows of a 2-D destination grid are walked in ty x tx tiles
//                  (1-D: natural order); ty * tx consecutive work slots = one
//                  patch
//   radix sort     rows by key -> row of every slot; scatter -> slot of a row
//   slot_lengths + scan   entries per slot -> patch_rowptr (patch-major CSR)
//   patch_entry_keys      per CSR entry: key = patch * n_a + col, payload =
//                  its position q in slot order; patch_val[q] = S
//   radix sort     (key, q); head flags + scan -> index of the distinct
//                  (patch, col) pair
//   patch_bounds   per patch: lower_bound -> patch_ptr; per distinct pair:
//                  patch_ucol
//   patch_local    patch_lidx[q] = index of the entry's column in ITS patch's
//                  list; umax / emax by atomicMax
// ===========================================================================
namespace remap {
namespace {

struct PatchLayout {
    size_t row_keys_in, row_keys_out, rows_in, rid, slot_of_row, lens, keys_in,
        keys_out, q_in, q_out, head, uidx, temp, total;
    size_t temp_bytes;
};

int patch_layout(int64_t n_rows, int64_t nnz, PatchLayout *lay)
{
    const size_t nr = static_cast<size_t>(n_rows > 0 ? n_rows : 1);
    const size_t ne = static_cast<size_t>(nnz > 0 ? nnz : 1);
    size_t a = 0, b = 0, c = 0, d = 0;
    REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
        nullptr, a, static_cast<const uint64_t *>(nullptr),
        static_cast<uint64_t *>(nullptr), static_cast<const int32_t *>(nullptr),
        static_cast<int32_t *>(nullptr), nr, 0u, 64u)));
    REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
        nullptr, b, static_cast<const uint64_t *>(nullptr),
        static_cast<uint64_t *>(nullptr),
        static_cast<const uint32_t *>(nullptr),
        static_cast<uint32_t *>(nullptr), ne, 0u, 64u)));
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        nullptr, c, static_cast<const uint32_t *>(nullptr),
        static_cast<uint32_t *>(nullptr), 0u, ne, rocprim::plus<uint32_t>())));
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        nullptr, d, static_cast<const int32_t *>(nullptr),
        static_cast<int32_t *>(nullptr), 0, nr + 1, rocprim::plus<int32_t>())));
    lay->temp_bytes = a;
    if (b > lay->temp_bytes) lay->temp_bytes = b;
    if (c > lay->temp_bytes) lay->temp_bytes = c;
    if (d > lay->temp_bytes) lay->temp_bytes = d;
    size_t off = 0;
    lay->row_keys_in = off;  off += align_g(nr * 8);
    lay->row_keys_out = off; off += align_g(nr * 8);
    lay->rows_in = off;      off += align_g(nr * 4);
    lay->rid = off;          off += align_g(nr * 4);
    lay->slot_of_row = off;  off += align_g(nr * 4);
    lay->lens = off;         off += align_g((nr + 1) * 4);
    lay->keys_in = off;      off += align_g(ne * 8);
    lay->keys_out = off;     off += align_g(ne * 8);
    lay->q_in = off;         off += align_g(ne * 4);
    lay->q_out = off;        off += align_g(ne * 4);
    lay->head = off;         off += align_g(ne * 4);
    lay->uidx = off;         off += align_g(ne * 4);
    lay->temp = off;         off += align_g(lay->temp_bytes);
    lay->total = off;
    return REMAP_OK;
}

__global__ __launch_bounds__(kBlock) void tile_keys(
    int64_t n_rows, int64_t row_offset, int64_t mx, int64_t ty, int64_t tx,
    uint64_t *__restrict__ keys, int32_t *__restrict__ rows)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_rows)
        return;
    const int64_t i = row_offset + r;
    const int64_t jy = i / mx;
    const int64_t jx = i - jy * mx;
    const int64_t ntx = (mx + tx - 1) / tx;
    keys[r] = static_cast<uint64_t>(
        ((jy / ty) * ntx + jx / tx) * (ty * tx) + (jy % ty) * tx + jx % tx);
    rows[r] = static_cast<int32_t>(r);
}

__global__ __launch_bounds__(kBlock) void slot_lengths(
    int64_t n_rows, const int32_t *__restrict__ rid,
    const int64_t *__restrict__ rowptr, int32_t *__restrict__ lens)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s > n_rows)
        return;
    lens[s] = s < n_rows
                  ? static_cast<int32_t>(rowptr[rid[s] + 1] - rowptr[rid[s]])
                  : 0;
}

__global__ __launch_bounds__(kBlock) void patch_entry_keys(
    int64_t n_rows, int64_t n_a, int64_t patch_rows,
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const int32_t *__restrict__ slot_of,
    const int32_t *__restrict__ prow, uint64_t *__restrict__ keys,
    uint32_t *__restrict__ q_of, double *__restrict__ pval)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_rows)
        return;
    const int64_t slot = slot_of[r];
    const uint64_t patch = static_cast<uint64_t>(slot / patch_rows);
    const int64_t q0 = prow[slot];
    const int64_t e0 = rowptr[r];
    for (int64_t e = e0; e < rowptr[r + 1]; ++e) {
        const int64_t q = q0 + (e - e0);
        keys[e] = patch * static_cast<uint64_t>(n_a) +
                  static_cast<uint64_t>(col[e]);
        q_of[e] = static_cast<uint32_t>(q);
        pval[q] = val[e];
    }
}

__global__ __launch_bounds__(kBlock) void flag_pair_heads(
    int64_t nnz, const uint64_t *__restrict__ keys,
    uint32_t *__restrict__ head)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz)
        return;
    head[n] = (n == 0 || keys[n] != keys[n - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(kBlock) void patch_bounds(
    int64_t n_patches, int64_t nnz, int64_t n_a, int64_t n_rows,
    int64_t patch_rows, const uint64_t *__restrict__ keys,
    const uint32_t *__restrict__ head, const uint32_t *__restrict__ uidx,
    const int32_t *__restrict__ prow, int32_t *__restrict__ pptr,
    int64_t *__restrict__ stats)
{
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p > n_patches)
        return;
    auto first_of = [&](int64_t patch) {
        const uint64_t first =
            static_cast<uint64_t>(patch) * static_cast<uint64_t>(n_a);
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < first)
                lo = mid + 1;
            else
                hi = mid;
        }
        const int64_t n_union =
            nnz > 0 ? (int64_t)uidx[nnz - 1] + (int64_t)head[nnz - 1] : 0;
        return lo < nnz ? (int64_t)uidx[lo] : n_union;
    };
    const int64_t mine = first_of(p);
    pptr[p] = static_cast<int32_t>(mine);
    if (p == n_patches) {
        stats[0] = mine;   // distinct (patch, col) pairs
        return;
    }
    const int64_t next = first_of(p + 1);
    atomicMax(reinterpret_cast<unsigned long long *>(stats + 1),
              static_cast<unsigned long long>(next - mine));
    int64_t s1 = (p + 1) * patch_rows;
    if (s1 > n_rows)
        s1 = n_rows;
    atomicMax(reinterpret_cast<unsigned long long *>(stats + 2),
              static_cast<unsigned long long>(prow[s1] - prow[p * patch_rows]));
}

__global__ __launch_bounds__(kBlock) void patch_local(
    int64_t nnz, int64_t n_a, const uint64_t *__restrict__ keys,
    const uint32_t *__restrict__ q_sorted, const uint32_t *__restrict__ head,
    const uint32_t *__restrict__ uidx, const int32_t *__restrict__ pptr,
    int32_t *__restrict__ ucol, int32_t *__restrict__ lidx)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz)
        return;
    const uint64_t key = keys[n];
    const uint32_t u = head[n] ? uidx[n] : uidx[n] - 1u;
    const uint64_t patch = key / static_cast<uint64_t>(n_a);
    if (head[n])
        ucol[u] = static_cast<int32_t>(key % static_cast<uint64_t>(n_a));
    lidx[q_sorted[n]] = static_cast<int32_t>(u) - pptr[patch];
}

}  // namespace

int patches_workspace(int64_t n_rows, int64_t nnz, size_t *bytes_out)
{
    if (!bytes_out || n_rows < 0 || nnz < 0)
        return fail(REMAP_ERR_ARG, "remap_patches_workspace: bad args");
    PatchLayout lay;
    const int rc = patch_layout(n_rows, nnz, &lay);
    if (rc != REMAP_OK)
        return rc;
    *bytes_out = lay.total;
    return REMAP_OK;
}

int patches_build(const remap_csr *A, const int64_t *grid_dims,
                  int64_t row_offset, int32_t tile_y, int32_t tile_x,
                  int32_t *row_order_out, int32_t *pptr, int32_t *ucol,
                  int32_t *prow, int32_t *lidx, double *pval, int64_t *stats,
                  void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    if (!A || !pptr || !ucol || !prow || !lidx || !pval || !stats)
        return fail(REMAP_ERR_ARG, "remap_patches_build: NULL argument");
    const int64_t n_rows = A->n_rows, nnz = A->nnz, n_a = A->n_cols;
    if (n_rows <= 0 || nnz <= 0 || n_a <= 0)
        return fail(REMAP_ERR_ARG,
                    "remap_patches_build: an empty matrix has no plan");
    if (!A->rowptr || !A->col || !A->val)
        return fail(REMAP_ERR_ARG, "remap_patches_build: NULL CSR array");
    if (tile_y < 1 || tile_x < 1)
        return fail(REMAP_ERR_ARG, "remap_patches_build: tile %d x %d",
                    tile_y, tile_x);
    if (grid_dims && !row_order_out)
        return fail(REMAP_ERR_ARG,
                    "remap_patches_build: a 2-D walk needs row_order_out");
    const int64_t patch_rows = (int64_t)tile_y * tile_x;
    const int64_t n_patches = (n_rows + patch_rows - 1) / patch_rows;
    if (nnz >= (int64_t(1) << 31) ||
        static_cast<long double>(n_patches) * n_a >= 9.0e18L)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_patches_build: mapping too large for 32-bit "
                    "patch offsets");
    if (grid_dims &&
        (grid_dims[0] <= 0 || grid_dims[1] <= 0 || row_offset < 0 ||
         row_offset + n_rows > grid_dims[0] * grid_dims[1]))
        return fail(REMAP_ERR_ARG,
                    "remap_patches_build: rows outside the grid");
    PatchLayout lay;
    int rc = patch_layout(n_rows, nnz, &lay);
    if (rc != REMAP_OK)
        return rc;
    if (!workspace || workspace_bytes < lay.total)
        return fail(REMAP_ERR_WORKSPACE,
                    "remap_patches_build: workspace of %zu bytes, need %zu",
                    workspace_bytes, lay.total);
    char *ws = static_cast<char *>(workspace);
    uint64_t *rk_in = reinterpret_cast<uint64_t *>(ws + lay.row_keys_in);
    uint64_t *rk_out = reinterpret_cast<uint64_t *>(ws + lay.row_keys_out);
    int32_t *rows_in = reinterpret_cast<int32_t *>(ws + lay.rows_in);
    int32_t *rid = reinterpret_cast<int32_t *>(ws + lay.rid);
    int32_t *slot_of = reinterpret_cast<int32_t *>(ws + lay.slot_of_row);
    int32_t *lens = reinterpret_cast<int32_t *>(ws + lay.lens);
    uint64_t *k_in = reinterpret_cast<uint64_t *>(ws + lay.keys_in);
    uint64_t *k_out = reinterpret_cast<uint64_t *>(ws + lay.keys_out);
    uint32_t *q_in = reinterpret_cast<uint32_t *>(ws + lay.q_in);
    uint32_t *q_out = reinterpret_cast<uint32_t *>(ws + lay.q_out);
    uint32_t *head = reinterpret_cast<uint32_t *>(ws + lay.head);
    uint32_t *uidx = reinterpret_cast<uint32_t *>(ws + lay.uidx);
    void *temp = ws + lay.temp;

    REMAP_HIP_CHECK(hipMemsetAsync(stats, 0, 3 * sizeof(int64_t), stream));
    if (grid_dims) {
        hipLaunchKernelGGL(tile_keys, dim3(blocks_for(n_rows)), dim3(kBlock),
                           0, stream, n_rows, row_offset, grid_dims[1],
                           (int64_t)tile_y, (int64_t)tile_x, rk_in, rows_in);
        REMAP_HIP_CHECK(hipGetLastError());
        size_t tb = lay.temp_bytes;
        REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
            temp, tb, static_cast<const uint64_t *>(rk_in), rk_out,
            static_cast<const int32_t *>(rows_in), rid,
            static_cast<size_t>(n_rows), 0u, 64u, stream)));
        hipLaunchKernelGGL(invert_order, dim3(blocks_for(n_rows)),
                           dim3(kBlock), 0, stream, n_rows, rid, slot_of,
                           row_order_out);
        REMAP_HIP_CHECK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(identity_order, dim3(blocks_for(n_rows)),
                           dim3(kBlock), 0, stream, n_rows, rid, slot_of);
        REMAP_HIP_CHECK(hipGetLastError());
    }
    // entries per slot -> patch_rowptr
    hipLaunchKernelGGL(slot_lengths, dim3(blocks_for(n_rows + 1)),
                       dim3(kBlock), 0, stream, n_rows, rid, A->rowptr, lens);
    REMAP_HIP_CHECK(hipGetLastError());
    size_t tb = lay.temp_bytes;
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        temp, tb, static_cast<const int32_t *>(lens), prow, 0,
        static_cast<size_t>(n_rows + 1), rocprim::plus<int32_t>(), stream)));
    // distinct (patch, col) pairs
    hipLaunchKernelGGL(patch_entry_keys, dim3(blocks_for(n_rows)),
                       dim3(kBlock), 0, stream, n_rows, n_a, patch_rows,
                       A->rowptr, A->col, A->val, slot_of, prow, k_in, q_in,
                       pval);
    REMAP_HIP_CHECK(hipGetLastError());
    tb = lay.temp_bytes;
    REMAP_HIP_CHECK((rocprim::radix_sort_pairs(
        temp, tb, static_cast<const uint64_t *>(k_in), k_out,
        static_cast<const uint32_t *>(q_in), q_out, static_cast<size_t>(nnz),
        0u, 64u, stream)));
    hipLaunchKernelGGL(flag_pair_heads, dim3(blocks_for(nnz)), dim3(kBlock),
                       0, stream, nnz, k_out, head);
    REMAP_HIP_CHECK(hipGetLastError());
    tb = lay.temp_bytes;
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        temp, tb, static_cast<const uint32_t *>(head), uidx, 0u,
        static_cast<size_t>(nnz), rocprim::plus<uint32_t>(), stream)));
    hipLaunchKernelGGL(patch_bounds, dim3(blocks_for(n_patches + 1)),
                       dim3(kBlock), 0, stream, n_patches, nnz, n_a, n_rows,
                       patch_rows, k_out, head, uidx, prow, pptr, stats);
    REMAP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(patch_local, dim3(blocks_for(nnz)), dim3(kBlock), 0,
                       stream, nnz, n_a, k_out, q_out, head, uidx, pptr, ucol,
                       lidx);
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

}  // namespace remap

extern "C" {

int remap_patches_workspace(int64_t n_rows, int64_t nnz, size_t *bytes_out)
{
    return remap::patches_workspace(n_rows, nnz, bytes_out);
}

int remap_patches_build(const remap_csr *A, const int64_t *grid_dims,
                        int64_t row_offset, int32_t tile_y, int32_t tile_x,
                        int32_t *row_order_out, int32_t *patch_ptr,
                        int32_t *patch_ucol, int32_t *patch_rowptr,
                        int32_t *patch_lidx, double *patch_val,
                        int64_t *stats_out, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    return remap::patches_build(A, grid_dims, row_offset, tile_y, tile_x,
                                row_order_out, patch_ptr, patch_ucol,
                                patch_rowptr, patch_lidx, patch_val,
                                stats_out, workspace, workspace_bytes,
                                static_cast<hipStream_t>(stream));
}

}  // extern "C"

// ===========================================================================
// remap_schedule_auto: WHICH schedule a mapping gets, decided and built in
// one call.  The rules are measurements:
//
//   1. LDS patches pay when neighbouring destination rows share most of
//      their source rows and rows are short (bilinear, coarse -> fine:
//      BASELINE config 4, 2.2x): the largest tile of kAutoTiles whose LDS
//      image fits 100 KB (1 KiB per staged row first, then 512 B), kept if
//      distinct / entries <= 0.30.  Never for entry-rich rows (>= 10 entries
//      per non-empty row: the compute phase is LDS-issue-bound there).
//   2. Else row groups when rows share columns at all (union / entries <=
//      0.95): 2 x 4 groups inside 32 x 32 supertiles for entry-rich rows,
//      2 K-tiles per wave (1 in masked mode), one group per wave, single-
//      wave workgroups; otherwise 2 x 2 groups in row-major order, one
//      group per wave, 4-wave workgroups.
//   3. Else the plain wave-per-row kernel -- in 32 x 32 tile order for
//      entry-rich rows on a 2-D grid (keeps the stencil band in L2).
//
// The schedule's arrays are laid out in ONE caller-provided device arena;
// the returned struct carries the ready-to-copy remap_apply_args fields.
// ===========================================================================
namespace remap {
namespace {

constexpr int kAutoTiles[8][2] = {{24, 24}, {32, 16}, {24, 16}, {16, 16},
                                  {8, 16},  {8, 8},   {6, 8},   {4, 8}};
constexpr int kAutoTiles1D[2][2] = {{1, 256}, {1, 64}};
constexpr int64_t kAutoLdsBudget = 100 * 1024;
constexpr double kAutoPatchRatio = 0.30;
constexpr double kAutoGroupRatio = 0.95;

struct Arena {
    char *base;
    size_t size, used;
    template <typename T>
    T *take(size_t n)
    {
        const size_t bytes = align_g(n * sizeof(T));
        if (used + bytes > size)
            return nullptr;
        T *p = reinterpret_cast<T *>(base + used);
        used += bytes;
        return p;
    }
};

size_t arena_need(int64_t n_rows, int64_t nnz)
{
    const size_t nr = static_cast<size_t>(n_rows), ne = static_cast<size_t>(nnz);
    const size_t order = align_g(nr * 4);
    const size_t patch = align_g((nr + 1) * 4) + align_g(ne * 4) +
                         align_g((nr + 1) * 4) + align_g(ne * 4) +
                         align_g(ne * 8);
    const size_t n_groups = nr / 4 + 2;
    const size_t n_super = nr / 32 + 2;
    const size_t group = align_g(2 * (n_groups + 1) * 8) +
                         2 * align_g((ne + 32) * 4) + align_g((ne + 128) * 8) +
                         align_g(n_groups * 8 * 4) + align_g(n_groups * 8 * 8) +
                         // the shared lists of entry-rich mappings
                         align_g(2 * (n_super + 1) * 8) +
                         2 * align_g((ne + 256) * 4);
    return order + (patch > group ? patch : group) + align_g(64);
}

__global__ __launch_bounds__(kBlock) void count_nonempty(
    int64_t n_rows, const int64_t *__restrict__ rowptr,
    unsigned long long *__restrict__ count)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool has = r < n_rows && rowptr[r + 1] > rowptr[r];
    const unsigned long long b = __builtin_amdgcn_ballot_w64(has);
    if ((threadIdx.x & (kWave - 1)) == 0 && b)
        atomicAdd(count, (unsigned long long)__builtin_popcountll(b));
}

int read_back(void *host, const void *dev, size_t bytes, hipStream_t stream)
{
    REMAP_HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost,
                                   stream));
    REMAP_HIP_CHECK(hipStreamSynchronize(stream));
    return REMAP_OK;
}

void set_tune(remap_schedule *s, int mode, int family, int vec, int tiles,
              int per_wave, int map = 0)
{
    s->tune[mode][0] = family;
    s->tune[mode][1] = vec;
    s->tune[mode][2] = tiles;
    s->tune[mode][3] = per_wave;
    s->tune[mode][4] = map;
}

}  // namespace

int schedule_sizes(int64_t n_rows, int64_t nnz, size_t *arena_bytes,
                   size_t *workspace_bytes)
{
    if (!arena_bytes || !workspace_bytes || n_rows < 0 || nnz < 0)
        return fail(REMAP_ERR_ARG, "remap_schedule_sizes: bad args");
    GroupLayout gl;
    PatchLayout pl;
    int rc = group_layout(n_rows, nnz, &gl);
    if (rc != REMAP_OK)
        return rc;
    rc = patch_layout(n_rows, nnz, &pl);
    if (rc != REMAP_OK)
        return rc;
    *workspace_bytes = gl.total > pl.total ? gl.total : pl.total;
    *arena_bytes = arena_need(n_rows > 0 ? n_rows : 1, nnz > 0 ? nnz : 1);
    return REMAP_OK;
}

int schedule_auto(const remap_csr *A, const double *frac_b,
                  const int64_t *grid_dims, int32_t n_dims,
                  int64_t row_offset, void *arena_ptr, size_t arena_bytes,
                  void *workspace, size_t workspace_bytes,
                  remap_schedule *out, hipStream_t stream)
{
    if (!A || !out)
        return fail(REMAP_ERR_ARG, "remap_schedule_auto: NULL argument");
    memset(out, 0, sizeof(*out));
    const int64_t n_rows = A->n_rows, nnz = A->nnz;
    if (n_dims < 0 || n_dims > 2 || (n_dims > 0 && !grid_dims))
        return fail(REMAP_ERR_ARG, "remap_schedule_auto: bad grid_dims");
    if (n_dims == 0 || nnz <= 0 || n_rows <= 0)
        return REMAP_OK;   // no destination grid / nothing to schedule
    if (!frac_b || !arena_ptr || !workspace)
        return fail(REMAP_ERR_ARG, "remap_schedule_auto: NULL buffer");
    if (arena_bytes < arena_need(n_rows, nnz))
        return fail(REMAP_ERR_WORKSPACE,
                    "remap_schedule_auto: arena of %zu bytes, need %zu",
                    arena_bytes, arena_need(n_rows, nnz));
    const bool two_d = n_dims == 2;
    const int64_t *dims2 = two_d ? grid_dims : nullptr;
    Arena arena{static_cast<char *>(arena_ptr), arena_bytes, 0};
    int64_t *stats = arena.take<int64_t>(8);
    int32_t *order = arena.take<int32_t>(n_rows);
    const size_t mark = arena.used;

    // entry-rich?
    REMAP_HIP_CHECK(hipMemsetAsync(stats, 0, 8 * sizeof(int64_t), stream));
    hipLaunchKernelGGL(count_nonempty, dim3(blocks_for(n_rows)), dim3(kBlock),
                       0, stream, n_rows, A->rowptr,
                       reinterpret_cast<unsigned long long *>(stats + 4));
    REMAP_HIP_CHECK(hipGetLastError());
    int64_t nonempty = 0;
    int rc = read_back(&nonempty, stats + 4, sizeof(int64_t), stream);
    if (rc != REMAP_OK)
        return rc;
    const bool entry_rich =
        nonempty > 0 && static_cast<double>(nnz) / nonempty >= 10.0;
    out->entry_rich = entry_rich ? 1 : 0;

    // 1. LDS patches
    if (!entry_rich) {
        const int (*tiles)[2] = two_d ? kAutoTiles : kAutoTiles1D;
        const int n_tiles = two_d ? 8 : 2;
        for (int row_bytes = 1024; row_bytes >= 512; row_bytes /= 2) {
            bool fits = false;
            for (int t = 0; t < n_tiles && !fits; ++t) {
                const int ty = tiles[t][0], tx = tiles[t][1];
                const int64_t rows = (int64_t)ty * tx;
                const int64_t n_patches = (n_rows + rows - 1) / rows;
                arena.used = mark;
                int32_t *pptr = arena.take<int32_t>(n_patches + 1);
                int32_t *ucol = arena.take<int32_t>(nnz);
                int32_t *prow = arena.take<int32_t>(n_rows + 1);
                int32_t *lidx = arena.take<int32_t>(nnz);
                double *pval = arena.take<double>(nnz);
                if (!pval)
                    return fail(REMAP_ERR_WORKSPACE,
                                "remap_schedule_auto: arena too small");
                rc = patches_build(A, dims2, row_offset, ty, tx,
                                   two_d ? order : nullptr, pptr, ucol, prow,
                                   lidx, pval, stats, workspace,
                                   workspace_bytes, stream);
                if (rc != REMAP_OK)
                    return rc;
                int64_t st3[3];
                rc = read_back(st3, stats, sizeof(st3), stream);
                if (rc != REMAP_OK)
                    return rc;
                const int64_t footprint = (st3[1] + 1) * row_bytes +
                                          st3[2] * 12 + rows * 24 + 32;
                if (footprint > kAutoLdsBudget)
                    continue;   // does not fit: the next, smaller tile
                fits = true;
                const double ratio = static_cast<double>(st3[0]) / nnz;
                if (ratio <= kAutoPatchRatio) {
                    out->family = 5;
                    out->row_order = two_d ? order : nullptr;
                    out->patch_ptr = pptr;
                    out->patch_ucol = ucol;
                    out->patch_rowptr = prow;
                    out->patch_lidx = lidx;
                    out->patch_val = pval;
                    out->patch_rows = static_cast<int32_t>(rows);
                    out->patch_umax = static_cast<int32_t>(st3[1]);
                    out->patch_emax = static_cast<int32_t>(st3[2]);
                    out->patch_row_bytes = row_bytes;
                    out->n_patches = n_patches;
                    out->tile_y = ty;
                    out->tile_x = tx;
                    out->ratio = ratio;
                    out->n_distinct = st3[0];
                    out->arena_used = arena.used;
                    for (int mode = 0; mode < 3; ++mode)
                        set_tune(out, mode, 5, 0, 0, 0);
                    return REMAP_OK;
                }
                // fits, too little reuse: smaller tiles share even less
            }
            if (fits)
                break;
        }
    }

    // 2. row groups
    {
        const int G = entry_rich ? 8 : 4;
        const int st = entry_rich ? 32 : 0;
        const int64_t n_groups = (n_rows + G - 1) / G;
        arena.used = mark;
        int64_t *meta = arena.take<int64_t>(2 * (n_groups + 1));
        int32_t *gcol = arena.take<int32_t>(nnz + 32);
        int32_t *gmask = arena.take<int32_t>(nnz + 32);
        double *gw = arena.take<double>(nnz + 128);
        int32_t *rid = arena.take<int32_t>(n_groups * G);
        double *gfrac = arena.take<double>(n_groups * G);
        if (!gfrac)
            return fail(REMAP_ERR_WORKSPACE,
                        "remap_schedule_auto: arena too small");
        // entry-rich: the group tiles nested in 4 x 8 tiles, the
        // supergroups of the shared form (spmm_groupshare.h)
        const int share = entry_rich ? 4 : 0;
        rc = groups_build(A, frac_b, G, dims2, row_offset, st, share,
                          two_d ? order : nullptr, meta, gcol, gmask, gw, rid,
                          gfrac, stats, workspace, workspace_bytes, stream);
        if (rc != REMAP_OK)
            return rc;
        int64_t n_union = 0;
        rc = read_back(&n_union, stats, sizeof(int64_t), stream);
        if (rc != REMAP_OK)
            return rc;
        const double ratio = static_cast<double>(n_union) / nnz;
        int64_t *smeta = nullptr;
        int32_t *scol = nullptr, *smask = nullptr;
        int64_t n_share = 0;
        if (share && ratio <= kAutoGroupRatio) {
            // (the union arrays above stay as they are: the arena was sized
            // for both)
            const int64_t n_super = (n_rows + 8 * share - 1) / (8 * share);
            smeta = arena.take<int64_t>(2 * (n_super + 1));
            scol = arena.take<int32_t>(nnz + 256);
            smask = arena.take<int32_t>(nnz + 256);
            if (!smask)
                return fail(REMAP_ERR_WORKSPACE,
                            "remap_schedule_auto: arena too small");
            rc = share_build(A, rid, share, smeta, scol, smask, stats + 1,
                             workspace, workspace_bytes, stream);
            if (rc != REMAP_OK)
                return rc;
            rc = read_back(&n_share, stats + 1, sizeof(int64_t), stream);
            if (rc != REMAP_OK)
                return rc;
        }
        if (ratio <= kAutoGroupRatio) {
            out->family = 10;
            out->row_order = two_d ? order : nullptr;
            out->group_meta = meta;
            out->group_col = gcol;
            out->group_w = gw;
            out->group_mask = gmask;
            out->group_rid = rid;
            out->group_frac = gfrac;
            out->n_groups = n_groups;
            out->group_rows = G;
            out->super_tile = st;
            out->ratio = ratio;
            out->n_distinct = n_union;
            out->arena_used = arena.used;
            out->share_meta = smeta;
            out->share_col = scol;
            out->share_mask = smask;
            out->share_waves = smask ? share : 0;
            out->n_share_union = n_share;
            for (int mode = 0; mode < 3; ++mode) {
                // entry-rich: single-wave workgroups, one group each, the
                // K-chunks of a group side by side in the work list (its
                // schedule is fetched once per XCD: config 5 22.5 -> 22.0
                // ms, masked 27.9 -> 26.7); round 6: the frac_b and raw
                // modes in the shared form -- four waves, one union through
                // an LDS ring (config 5 21.0 -> 19.7 ms on one box; a call
                // it cannot serve takes the 8-row groups as before)
                if (entry_rich) {
                    set_tune(out, mode, 10, 1,
                             mode == REMAP_MODE_MASKED ? 1 : 2, 1, 3);
                    if (mode != REMAP_MODE_MASKED && smask) {
                        out->tune[mode][1] = 0;
                        out->tune[mode][3] = 0;
                        out->tune[mode][5] = 32;
                    }
                } else {
                    set_tune(out, mode, 10, 0, 0, 1);
                }
            }
            return REMAP_OK;
        }
    }

    // 3. plain kernels; entry-rich rows on a 2-D grid in 32 x 32 tile order
    arena.used = mark;
    if (two_d && entry_rich) {
        // only the order is wanted: a patch build with 32 x 32 tiles yields
        // it (its other outputs land in the arena and are abandoned)
        int32_t *pptr = arena.take<int32_t>(n_rows / 1024 + 2);
        int32_t *ucol = arena.take<int32_t>(nnz);
        int32_t *prow = arena.take<int32_t>(n_rows + 1);
        int32_t *lidx = arena.take<int32_t>(nnz);
        double *pval = arena.take<double>(nnz);
        if (!pval)
            return fail(REMAP_ERR_WORKSPACE,
                        "remap_schedule_auto: arena too small");
        rc = patches_build(A, dims2, row_offset, 32, 32, order, pptr, ucol,
                           prow, lidx, pval, stats, workspace,
                           workspace_bytes, stream);
        if (rc != REMAP_OK)
            return rc;
        REMAP_HIP_CHECK(hipStreamSynchronize(stream));
        out->family = 6;
        out->row_order = order;
        out->tile_y = out->tile_x = 32;
        out->arena_used = mark;
        for (int mode = 0; mode < 3; ++mode)
            set_tune(out, mode, 6, 0, 2, 4, 2);
    }
    return REMAP_OK;
}

}  // namespace remap

extern "C" {

int remap_schedule_sizes(int64_t n_rows, int64_t nnz, size_t *arena_bytes,
                         size_t *workspace_bytes)
{
    return remap::schedule_sizes(n_rows, nnz, arena_bytes, workspace_bytes);
}

int remap_schedule_auto(const remap_csr *A, const double *frac_b,
                        const int64_t *grid_dims, int32_t n_dims,
                        int64_t row_offset, void *arena, size_t arena_bytes,
                        void *workspace, size_t workspace_bytes,
                        remap_schedule *schedule_out, void *stream)
{
    return remap::schedule_auto(A, frac_b, grid_dims, n_dims, row_offset,
                                arena, arena_bytes, workspace,
                                workspace_bytes, schedule_out,
                                static_cast<hipStream_t>(stream));
}

}  // extern "C"
