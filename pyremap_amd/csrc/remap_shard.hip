// remap_shard.hip -- the two device steps a ROW SHARD needs beside the apply
// kernels ("send each rank only the X rows in
// unique(col[shard])"):
//
//   remap_pack_columns   one-off per shard: the sorted list of DISTINCT source
//                        rows the shard's entries reference, and the shard's
//                        column indices renumbered into that compact space.
//                        The renumbering is monotone, so a row's entries keep
//                        their order -- the order the sums are taken in
//                        (remap_numpy.py:264-268 via scipy's csr_matvecs) --
//                        and the packed shard gives the same bits.
//   remap_gather_rows    per batch, on the rank (or device) that holds the
//                        field: out[b][i][:] = X[b][rows[i]][:], the packed
//                        buffer that travels to the shard's GPU.
//
// Works for ANY source-cell numbering: a (min, max) band of source rows only
// helps when the numbering follows the destination raster, which no MPAS
// mesh's does (tests/golden/qu240_cells.npz: one 1-degree destination row
// meets ids from 77-99 % of the id range).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "remap_common.h"

namespace remap {
namespace {

constexpr size_t kAlignShard = 256;

size_t align_up_shard(size_t n)
{
    return (n + kAlignShard - 1) / kAlignShard * kAlignShard;
}

struct PackLayout {
    size_t flag, slot, temp, total, temp_bytes;
};

int pack_layout(int64_t n_cols, PackLayout *lay)
{
    const size_t n = static_cast<size_t>(n_cols > 0 ? n_cols : 1) + 1;
    size_t scan_bytes = 0;
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        nullptr, scan_bytes, static_cast<const uint32_t *>(nullptr),
        static_cast<uint32_t *>(nullptr), 0u, n, rocprim::plus<uint32_t>())));
    lay->temp_bytes = scan_bytes;
    size_t off = 0;
    lay->flag = off; off += align_up_shard(n * 4);
    lay->slot = off; off += align_up_shard(n * 4);
    lay->temp = off; off += align_up_shard(scan_bytes);
    lay->total = off;
    return REMAP_OK;
}

__global__ __launch_bounds__(kBlock) void mark_columns(
    int64_t nnz, int64_t n_cols, const int32_t *__restrict__ col,
    uint32_t *__restrict__ flag, int64_t *__restrict__ bad)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz)
        return;
    const int64_t c = col[n];
    if (c < 0 || c >= n_cols) {
        atomicAdd(reinterpret_cast<unsigned long long *>(bad), 1ull);
        return;
    }
    flag[c] = 1u;   // benign race: every writer stores the same value
}

__global__ __launch_bounds__(kBlock) void list_columns(
    int64_t n_cols, const uint32_t *__restrict__ flag,
    const uint32_t *__restrict__ slot, int32_t *__restrict__ ucols,
    int64_t *__restrict__ n_out)
{
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c > n_cols)
        return;
    if (c == n_cols) {
        *n_out = slot[n_cols];    // the scan ran over n_cols + 1 flags
        return;
    }
    if (flag[c])
        ucols[slot[c]] = static_cast<int32_t>(c);
}

__global__ __launch_bounds__(kBlock) void renumber_columns(
    int64_t nnz, int64_t n_cols, const int32_t *__restrict__ col,
    const uint32_t *__restrict__ slot, int32_t *__restrict__ col_out)
{
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= nnz)
        return;
    const int64_t c = col[n];
    col_out[n] = (c >= 0 && c < n_cols) ? static_cast<int32_t>(slot[c]) : 0;
}

// `lpr` lanes (a power of two <= 64) per (batch, listed row, piece): UNIT
// bytes per lane, so long rows move in 1 KiB pieces per wave and short ones
// -- (Time, nCells): 8 bytes per row -- share a wave.
template <typename UNIT>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(
    const char *__restrict__ src, int64_t src_batch_stride,
    int64_t src_row_stride, const int32_t *__restrict__ rows, int64_t n_rows,
    int64_t n_batch, int64_t row_bytes, int64_t pieces, int32_t lpr_shift,
    char *__restrict__ dst)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t item = gid >> lpr_shift;
    const int64_t l = gid & ((int64_t(1) << lpr_shift) - 1);
    if (item >= n_batch * n_rows * pieces)
        return;
    const int64_t piece = item % pieces;
    const int64_t i = (item / pieces) % n_rows;
    const int64_t b = item / (pieces * n_rows);
    const int64_t off =
        ((piece << lpr_shift) + l) * (int64_t)sizeof(UNIT);
    if (off >= row_bytes)
        return;
    const int64_t r = rows[i];
    const UNIT v = *reinterpret_cast<const UNIT *>(
        src + b * src_batch_stride + r * src_row_stride + off);
    *reinterpret_cast<UNIT *>(dst + (b * n_rows + i) * row_bytes + off) = v;
}

bool aligned_to(const void *p, size_t a)
{
    return (reinterpret_cast<uintptr_t>(p) % a) == 0;
}

typedef unsigned int unit16 __attribute__((ext_vector_type(4)));

}  // namespace

int pack_columns_workspace(int64_t n_cols, size_t *bytes_out)
{
    if (!bytes_out || n_cols < 0)
        return fail(REMAP_ERR_ARG, "remap_pack_columns_workspace: bad args");
    PackLayout lay;
    const int rc = pack_layout(n_cols, &lay);
    if (rc != REMAP_OK)
        return rc;
    *bytes_out = lay.total;
    return REMAP_OK;
}

int pack_columns(const int32_t *col, int64_t nnz, int64_t n_cols,
                 int32_t *col_out, int32_t *ucols_out, int64_t *n_ucols_out,
                 int64_t *bad_out, void *workspace, size_t workspace_bytes,
                 hipStream_t stream)
{
    if (nnz < 0 || n_cols < 0 || n_cols >= (int64_t(1) << 31))
        return fail(REMAP_ERR_ARG, "remap_pack_columns: bad size");
    if (!n_ucols_out || !bad_out || (nnz > 0 && (!col || !col_out)) ||
        (n_cols > 0 && !ucols_out))
        return fail(REMAP_ERR_ARG, "remap_pack_columns: NULL pointer");
    PackLayout lay;
    const int rc = pack_layout(n_cols, &lay);
    if (rc != REMAP_OK)
        return rc;
    if (!workspace || workspace_bytes < lay.total)
        return fail(REMAP_ERR_WORKSPACE,
                    "remap_pack_columns: workspace of %zu bytes, need %zu",
                    workspace_bytes, lay.total);
    char *ws = static_cast<char *>(workspace);
    uint32_t *flag = reinterpret_cast<uint32_t *>(ws + lay.flag);
    uint32_t *slot = reinterpret_cast<uint32_t *>(ws + lay.slot);
    const size_t n = static_cast<size_t>(n_cols) + 1;
    REMAP_HIP_CHECK(hipMemsetAsync(flag, 0, n * 4, stream));
    REMAP_HIP_CHECK(hipMemsetAsync(bad_out, 0, sizeof(int64_t), stream));
    const uint32_t eblk = static_cast<uint32_t>((nnz + kBlock - 1) / kBlock);
    if (nnz > 0) {
        hipLaunchKernelGGL(mark_columns, dim3(eblk), dim3(kBlock), 0, stream,
                           nnz, n_cols, col, flag, bad_out);
        REMAP_HIP_CHECK(hipGetLastError());
    }
    size_t tb = lay.temp_bytes;
    REMAP_HIP_CHECK((rocprim::exclusive_scan(
        ws + lay.temp, tb, static_cast<const uint32_t *>(flag), slot, 0u, n,
        rocprim::plus<uint32_t>(), stream)));
    const uint32_t cblk = static_cast<uint32_t>((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(list_columns, dim3(cblk), dim3(kBlock), 0, stream,
                       n_cols, flag, slot, ucols_out, n_ucols_out);
    REMAP_HIP_CHECK(hipGetLastError());
    if (nnz > 0) {
        hipLaunchKernelGGL(renumber_columns, dim3(eblk), dim3(kBlock), 0,
                           stream, nnz, n_cols, col, slot, col_out);
        REMAP_HIP_CHECK(hipGetLastError());
    }
    return REMAP_OK;
}

int gather_rows(const void *src, int64_t n_batch, int64_t src_batch_stride,
                int64_t src_row_stride, const int32_t *rows, int64_t n_rows,
                int64_t row_bytes, void *dst, hipStream_t stream)
{
    if (n_batch < 0 || n_rows < 0 || row_bytes < 0 || src_batch_stride < 0 ||
        src_row_stride < 0)
        return fail(REMAP_ERR_ARG, "remap_gather_rows: negative size");
    if (n_batch == 0 || n_rows == 0 || row_bytes == 0)
        return REMAP_OK;
    if (!src || !dst || !rows)
        return fail(REMAP_ERR_ARG, "remap_gather_rows: NULL pointer");
    // widest unit every address involved is a multiple of
    int unit = 16;
    while (unit > 1 &&
           (!aligned_to(src, unit) || !aligned_to(dst, unit) ||
            row_bytes % unit || src_row_stride % unit ||
            src_batch_stride % unit))
        unit >>= 1;
    if (unit == 2)
        unit = 1;
    int32_t lpr_shift = 6;   // lanes per piece: 64, fewer for short rows
    while (lpr_shift > 0 && (int64_t(unit) << (lpr_shift - 1)) >= row_bytes)
        --lpr_shift;
    const int64_t piece_bytes = int64_t(unit) << lpr_shift;
    const int64_t pieces = (row_bytes + piece_bytes - 1) / piece_bytes;
    const int64_t lanes = (n_batch * n_rows * pieces) << lpr_shift;
    const int64_t grid = (lanes + kBlock - 1) / kBlock;
    if (grid > 0x7fffffffLL)
        return fail(REMAP_ERR_UNSUPPORTED,
                    "remap_gather_rows: %lld blocks; split the call",
                    (long long)grid);
    const char *s = static_cast<const char *>(src);
    char *d = static_cast<char *>(dst);
#define REMAP_GATHER(UNIT)                                                   \
    hipLaunchKernelGGL(gather_rows_kernel<UNIT>, dim3((uint32_t)grid),       \
                       dim3(kBlock), 0, stream, s, src_batch_stride,         \
                       src_row_stride, rows, n_rows, n_batch, row_bytes,     \
                       pieces, lpr_shift, d)
    switch (unit) {
    case 16: REMAP_GATHER(unit16); break;
    case 8:  REMAP_GATHER(uint64_t); break;
    case 4:  REMAP_GATHER(uint32_t); break;
    default: REMAP_GATHER(uint8_t); break;
    }
#undef REMAP_GATHER
    REMAP_HIP_CHECK(hipGetLastError());
    return REMAP_OK;
}

}  // namespace remap

extern "C" {

int remap_pack_columns_workspace(int64_t n_cols, size_t *bytes_out)
{
    return remap::pack_columns_workspace(n_cols, bytes_out);
}

int remap_pack_columns(const int32_t *col, int64_t nnz, int64_t n_cols,
                       int32_t *col_out, int32_t *ucols_out,
                       int64_t *n_ucols_out, int64_t *bad_out, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    return remap::pack_columns(col, nnz, n_cols, col_out, ucols_out,
                               n_ucols_out, bad_out, workspace,
                               workspace_bytes,
                               static_cast<hipStream_t>(stream));
}

int remap_gather_rows(const void *src, int64_t n_batch,
                      int64_t src_batch_stride_bytes,
                      int64_t src_row_stride_bytes, const int32_t *rows,
                      int64_t n_rows, int64_t row_bytes, void *dst,
                      void *stream)
{
    return remap::gather_rows(src, n_batch, src_batch_stride_bytes,
                              src_row_stride_bytes, rows, n_rows, row_bytes,
                              dst, static_cast<hipStream_t>(stream));
}

}  // extern "C"
