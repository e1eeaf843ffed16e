// Shared helpers of libremap_hip.so (gfx950 only).
#ifndef REMAP_COMMON_H
#define REMAP_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "remap_hip.h"

namespace remap {

// thread-local message behind remap_last_error()
char *error_buffer();
constexpr int kErrorBufferSize = 512;

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), kErrorBufferSize, fmt, ap);
    va_end(ap);
    return code;
}

inline int hip_fail(hipError_t err, const char *what)
{
    return fail(REMAP_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
}

#define REMAP_HIP_CHECK(expr)                                   \
    do {                                                        \
        hipError_t err__ = (expr);                              \
        if (err__ != hipSuccess)                                \
            return ::remap::hip_fail(err__, #expr);             \
    } while (0)

constexpr int kWave = 64;        // gfx950 wavefront
constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kXcds = 8;         // MI355X: 8 XCDs, blocks dealt round-robin

}  // namespace remap

#endif  // REMAP_COMMON_H
