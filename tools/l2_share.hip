// l2_share.hip -- does an MI355X L2 serve the SECOND reader of a line that
// another workgroup of the same XCD asked for a moment ago?  (GPU box only.)
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/l2_share tools/l2_share.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/_build/l2_share
// Every region of 64 KiB is read by TWO workgroups: block b and block b + S
// (S = 8: same XCD, dispatched back to back; S = 1: different XCDs -- no
// shared L2, the control; S = 8 x 2048: one "generation" of blocks later).
// The second reader can be delayed by D microseconds.  FETCH_SIZE per launch
// against the 1 GiB of unique bytes says what the second read cost at the
// fabric: 1.0 x = served by L2, 2.0 x = fetched again.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                          \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));        \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

constexpr int kLoads = 16;                 // 16 x 4 KiB per block = 64 KiB
constexpr size_t kRegionU4 = kLoads * 256;  // u4 elements per region

__global__ __launch_bounds__(256) void k_share(const u4 *__restrict__ src,
                                               u4 *__restrict__ sink,
                                               int S, int delay_ticks)
{
    const int b = blockIdx.x;
    size_t region;
    bool second = false;
    if (S == 0) {
        region = b;
    } else {
        const int win = b / (2 * S);
        const int in = b - win * 2 * S;
        second = in >= S;
        region = (size_t)win * S + (second ? in - S : in);
    }
    if (second && delay_ticks > 0) {
        const unsigned long long t0 = wall_clock64();   // 100 MHz
        while (wall_clock64() - t0 < (unsigned long long)delay_ticks)
            __builtin_amdgcn_s_sleep(8);
    }
    const u4 *p = src + region * kRegionU4 + threadIdx.x;
    u4 v[kLoads];
#pragma unroll
    for (int j = 0; j < kLoads; ++j)
        v[j] = p[j * 256];
    u4 acc = v[0];
#pragma unroll
    for (int j = 1; j < kLoads; ++j)
        acc ^= v[j];
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u)   // never: keeps loads
        sink[threadIdx.x] = acc;
}

int main()
{
    const size_t regions = 16384;   // 1 GiB of unique bytes
    const size_t bytes = regions * kRegionU4 * sizeof(u4);
    u4 *src, *sink;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(src, 1, bytes));
    // a second buffer to flush L2 / MALL between launches
    u4 *flush;
    CHECK(hipMalloc(&flush, bytes));
    CHECK(hipMemset(flush, 2, bytes));
    struct Case {
        int S;
        int delay_us;
        const char *what;
    };
    const Case cases[] = {
        {0, 0, "unique: every region read once"},
        {1, 0, "pairs on DIFFERENT XCDs (control, expect 2x)"},
        {8, 0, "pairs on one XCD, back to back"},
        {8, 1, "second reader 1 us later"},
        {8, 2, "second reader 2 us later"},
        {8, 4, "second reader 4 us later"},
        {8, 8, "second reader 8 us later"},
        {8, 16, "second reader 16 us later"},
        {8, 32, "second reader 32 us later"},
        {64, 0, "partner 8 blocks later on the XCD"},
        {512, 0, "partner 64 blocks later on the XCD"},
        {4096, 0, "partner 512 blocks later on the XCD"},
        {16384, 0, "partner 2048 blocks later on the XCD"},
    };
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (const Case &c : cases) {
        const int blocks = c.S == 0 ? (int)regions : (int)(2 * regions);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemsetAsync(flush, rep, bytes, 0));
            CHECK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(k_share, dim3(blocks), dim3(256), 0, 0, src,
                               sink, c.S, c.delay_us * 100);
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best)
                best = ms;
        }
        printf("CASE S=%d delay=%dus %-48s %.3f ms\n", c.S, c.delay_us,
               c.what, best);
    }
    return 0;
}
