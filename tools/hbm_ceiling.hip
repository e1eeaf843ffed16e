// hbm_ceiling.hip -- what this MI355X box sustains for plain streaming
// traffic, to put the remap kernel's numbers in context (GPU box only).
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/hbm_ceiling tools/hbm_ceiling.hip
// Variants: read-only, write-only, copy (1:1), and a 1:1 mix in 4 KiB rows
// like the remap kernel's access shape; plain vs non-temporal; grid sizes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                          \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));        \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(u4 *__restrict__ dst,
                                              const u4 *__restrict__ src,
                                              size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += stride) {
        if (NT)
            __builtin_nontemporal_store(__builtin_nontemporal_load(src + i),
                                        dst + i);
        else
            dst[i] = src[i];
    }
}

// 4 independent loads per thread before the stores
template <bool NT>
__global__ __launch_bounds__(256) void k_copy4(u4 *__restrict__ dst,
                                               const u4 *__restrict__ src,
                                               size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u4 a = src[i], b = src[i + stride], c = src[i + 2 * stride],
           d = src[i + 3 * stride];
        if (NT) {
            __builtin_nontemporal_store(a, dst + i);
            __builtin_nontemporal_store(b, dst + i + stride);
            __builtin_nontemporal_store(c, dst + i + 2 * stride);
            __builtin_nontemporal_store(d, dst + i + 3 * stride);
        } else {
            dst[i] = a;
            dst[i + stride] = b;
            dst[i + 2 * stride] = c;
            dst[i + 3 * stride] = d;
        }
    }
    for (; i < n; i += stride)
        dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_read(const u4 *__restrict__ src,
                                              size_t n, u4 *sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    u4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u4 a = src[i], b = src[i + stride], c = src[i + 2 * stride],
           d = src[i + 3 * stride];
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n; i += stride)
        acc ^= src[i];
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u)
        *sink = acc;
}

// non-persistent read: one block per 16 KiB (4 x 16 B per thread)
__global__ __launch_bounds__(256) void k_read_flat(const u4 *__restrict__ src,
                                                   size_t n, u4 *sink)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 >= n)
        return;
    u4 acc = src[i] ^ src[i + 256] ^ src[i + 512] ^ src[i + 768];
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u)
        *sink = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void k_write(u4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const u4 v = {1, 2, 3, (unsigned)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += stride) {
        if (NT)
            __builtin_nontemporal_store(v, dst + i);
        else
            dst[i] = v;
    }
}

// wave-per-4KiB-row copy: each wave reads one 4 KiB row (4 x 1 KiB loads) and
// writes one 4 KiB row, rows taken in order -- the remap kernel's shape
// without the gather
__global__ __launch_bounds__(256) void k_rowcopy(u4 *__restrict__ dst,
                                                 const u4 *__restrict__ src,
                                                 size_t nrows)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nw = ((size_t)gridDim.x * 256) >> 6;
    for (size_t r = wave; r < nrows; r += nw) {
        const u4 *s = src + r * 256 + lane;
        u4 a = s[0], b = s[64], c = s[128], d = s[192];
        u4 *o = dst + r * 256 + lane;
        __builtin_nontemporal_store(a, o);
        __builtin_nontemporal_store(b, o + 64);
        __builtin_nontemporal_store(c, o + 128);
        __builtin_nontemporal_store(d, o + 192);
    }
}

template <typename F>
double time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i)
        launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i)
        launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? atoll(argv[1]) : 1024) << 20;
    const size_t n = bytes / 16;
    u4 *src, *dst, *sink;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMalloc(&sink, 16));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMemset(dst, 2, bytes));
    printf("buffer %zu MiB each\n", bytes >> 20);
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32, 0};
    for (int g : grids) {
        const unsigned grid = g ? g : (unsigned)((n + 255) / 256);
        const double gb = bytes / 1e9;
        double t;
        t = time_ms([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, dst, src, n); }, 10);
        printf("grid %8u copy plain   %7.1f GB/s (r+w)\n", grid, 2 * gb / (t * 1e-3));
        t = time_ms([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, dst, src, n); }, 10);
        printf("grid %8u copy nt      %7.1f GB/s (r+w)\n", grid, 2 * gb / (t * 1e-3));
        if (g) {
            t = time_ms([&] { hipLaunchKernelGGL(k_copy4<false>, dim3(grid), dim3(256), 0, 0, dst, src, n); }, 10);
            printf("grid %8u copy4 plain  %7.1f GB/s (r+w)\n", grid, 2 * gb / (t * 1e-3));
            t = time_ms([&] { hipLaunchKernelGGL(k_copy4<true>, dim3(grid), dim3(256), 0, 0, dst, src, n); }, 10);
            printf("grid %8u copy4 nt     %7.1f GB/s (r+w)\n", grid, 2 * gb / (t * 1e-3));
            t = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, src, n, sink); }, 10);
            printf("grid %8u read         %7.1f GB/s\n", grid, gb / (t * 1e-3));
            t = time_ms([&] { hipLaunchKernelGGL(k_rowcopy, dim3(grid), dim3(256), 0, 0, dst, src, n / 256); }, 10);
            printf("grid %8u rowcopy 4KiB %7.1f GB/s (r+w)\n", grid, 2 * gb / (t * 1e-3));
        }
        if (!g) {
            t = time_ms([&] { hipLaunchKernelGGL(k_read_flat, dim3((unsigned)(n / 1024)), dim3(256), 0, 0, src, n, sink); }, 10);
            printf("grid %8u read flat    %7.1f GB/s\n", (unsigned)(n / 1024), gb / (t * 1e-3));
        }
        t = time_ms([&] { hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, dst, n); }, 10);
        printf("grid %8u write plain  %7.1f GB/s\n", grid, gb / (t * 1e-3));
        t = time_ms([&] { hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, dst, n); }, 10);
        printf("grid %8u write nt     %7.1f GB/s\n", grid, gb / (t * 1e-3));
    }
    double t = time_ms([&] { CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
    printf("hipMemcpy D2D          %7.1f GB/s (r+w)\n", 2 * (bytes / 1e9) / (t * 1e-3));
    return 0;
}
