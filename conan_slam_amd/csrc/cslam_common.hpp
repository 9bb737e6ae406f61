// cslam_common.hpp -- shared host-side helpers of libcslam_hip.so (error plumbing, small utilities).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/cslam.h"

namespace cslam
{

inline char* last_error_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define CSLAM_HIP_TRY(expr)                                                                              \
    do                                                                                                   \
    {                                                                                                    \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess)                                                                           \
        {                                                                                                \
            return ::cslam::fail(CSLAM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                                 __FILE__, __LINE__);                                                    \
        }                                                                                                \
    } while (0)

// Engines (single-filter handles and batched handles) alive in this process: kernels that wait inside a launch are only
// safe while ONE engine has the device to itself (see cslam_ekf.hip); every create / destroy counts here.
inline std::atomic<int>& live_engines()
{
    static std::atomic<int> n{0};
    return n;
}

inline int round_up(int v, int m)
{
    return ((v + m - 1) / m) * m;
}

// P, PHT and W1 all use leading dimensions that are multiples of this, so that every column starts
// 512-byte aligned and the downdate kernel can move whole 128-row tiles with 16-byte accesses and no
// bounds checks (padding rows of W1 are kept at zero, so padding of P never changes).
constexpr int kTile = 128;

} // namespace cslam
