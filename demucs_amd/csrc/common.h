// Shared helpers for the gfx950 engine: error plumbing, launch checks, device math.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/demucs_amd.h"

namespace mi {

// thread-local last-error text returned by mi_last_error()
char *last_error_buf();
int set_error(int code, const char *fmt, ...);

#define MI_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return mi::set_error(MI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                 __FILE__, __LINE__);                                             \
    } while (0)

#define MI_CHECK_LAUNCH() MI_HIP(hipGetLastError())

#define MI_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return mi::set_error(MI_EINVAL, __VA_ARGS__); \
    } while (0)

#define MI_TRY(expr)              \
    do {                          \
        int _r = (expr);          \
        if (_r != MI_OK) return _r; \
    } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// exact GELU, F.gelu default (erf form)
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// number of partial-sum slots per statistics row (spreads fp64 atomics over addresses)
constexpr int kStatSlots = 32;

}  // namespace mi
