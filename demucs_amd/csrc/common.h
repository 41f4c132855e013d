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

// Debug aid (tools/micro/poison_all.py): a hook called after EVERY kernel launch of the engine with the launch stream, so that
// a test can interleave a kernel that overwrites all VGPRs / AGPRs / LDS of the chip and show whether any engine kernel
// consumes state it did not write (what a co-resident process of another application would leave behind).
typedef void (*post_launch_hook_t)(void *stream);
extern post_launch_hook_t g_post_launch_hook;
extern int g_last_conv_route;          // mi_debug_last_conv_route (include/demucs_amd.h)

#define MI_CHECK_LAUNCH()                                                   \
    do {                                                                    \
        MI_HIP(hipGetLastError());                                          \
        if (mi::g_post_launch_hook) mi::g_post_launch_hook((void *)st);     \
    } while (0)

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

// exact GELU, F.gelu default (erf form): 0.5 x (1 + erf(x / sqrt 2)) with erf(t) = 1 - 2^(t Q(t)) on t = |x| <= 5.586
// (beyond it erfc < 2.4e-8).  Q is a degree-8 weighted least-squares fit of log2(erfc(t)) / t (tools/micro/fit_gelu.py);
// in float32 the result is within 1.1e-7 |x| of the float64 GELU everywhere (libm's erff form: 3.6e-7 |x| on the host,
// and ~50 instructions with a divergent branch against 16 here).
__device__ __forceinline__ float gelu_exact(float x) {
    const float t = fminf(fabsf(x), 5.586143494f);
    float q = 5.128879366e-07f;
    q = fmaf(q, t, -9.561274965e-06f);
    q = fmaf(q, t, 7.498410559e-05f);
    q = fmaf(q, t, -2.843918046e-04f);
    q = fmaf(q, t, 1.508691730e-05f);
    q = fmaf(q, t, 6.931011099e-03f);
    q = fmaf(q, t, -5.243470520e-02f);
    q = fmaf(q, t, -4.592214525e-01f);
    q = fmaf(q, t, -1.151104212e+00f);
    const float e = __builtin_amdgcn_exp2f(q * t);          // v_exp_f32; the argument lies in [-25.3, 0]
    return 0.5f * x * (1.0f + copysignf(1.0f - e, x));
}
// 1 / (1 + e^-x) as v_exp_f32 + v_rcp_f32 (each within 1 ulp: relative error < 3e-7, overflow -> 0 / 1 exactly): 4 instructions
// instead of ~20 for libm's expf and an IEEE division -- in the fused DConv kernels the GLU epilogue cost more VALU slots than
// the 1x1's multiply-adds
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

// LSTM cell update shared by the one-launch-per-step kernel (hkernels.hip) and the persistent kernel (lstm.hip).  The least
// significant mantissa bit of the returned h is replaced by `tag`: the persistent kernel's workgroups exchange h through memory
// as self-validating 4-byte values, the bit tells a value of step s from the buffer's previous occupant (step s - 2) -- at most one
// ulp, applied on both routes so that they agree bit for bit.  lstm_tag flips every second step and is 1 for steps 0 and 1 (a
// zero-filled buffer is then "not yet written").
// tanh as v_exp_f32 + v_rcp_f32: (1 - e) / (1 + e) with e = exp(-2 |x|) (each within 1 ulp; absolute error < 2e-7, exact +-1 for
// large |x|): ~8 instructions against libm tanhf's ~100 with divergent branches -- two of them sat on the critical path of every
// LSTM time step
__device__ __forceinline__ float tanh_f(float x) {
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(x));
    return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}
__device__ __forceinline__ unsigned lstm_tag(int step) { return (((unsigned)step >> 1) & 1u) ^ 1u; }
__device__ __forceinline__ float lstm_cell(float ai, float af, float ag, float ao, float &c, unsigned tag) {
    c = sigmoid_f(af) * c + sigmoid_f(ai) * tanh_f(ag);
    const float h = sigmoid_f(ao) * tanh_f(c);
    return __uint_as_float((__float_as_uint(h) & ~1u) | tag);
}

// two floats -> two bf16 / fp16 values (round to nearest even) in one dword, `a` in the low half
__device__ __forceinline__ unsigned pack_half2(int dtype, float a, float b) {
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 pk_f16x2 __attribute__((ext_vector_type(2)));
    if (dtype == 1 /* MI_DTYPE_BF16 */) {
        const pk_bf16x2 h = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, h);
    }
    const pk_f16x2 h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, h);
}

// One LDS-DMA wave instruction (global_load_lds_dwordx4: lane l's 16 bytes at `g` land at lds_wave_base + 16 l) issued from
// inline asm.  hipcc tracks the builtin form as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the next
// ds_read of the same array -- which drains a multi-stage ring right after its prefetch has been issued (found in the ISA of
// conv_gemm_half_img_kernel and attention_heads_kernel).  The asm form is invisible to that pass: ordering is then ONLY the
// kernel's own counted `s_waitcnt vmcnt(N)` + s_barrier (vmcnt counts these like any load).  lds_wave_base must be wave-uniform.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16(const void *g, void *lds_wave_base) {
    const unsigned l = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}
// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: a loop that walks a tensor advances the
// base with scalar instructions and spends no VALU slot on addresses
__device__ __forceinline__ void lds_dma16_s(const void *sbase, unsigned voff, void *lds_wave_base) {
    const unsigned l = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop

// packed fp32 math: v_pk_fma_f32 issues two IEEE fmas per lane per instruction (same results as two fmaf)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float x) { return (v2f){x, x}; }

// number of partial-sum slots per statistics row (spreads fp64 atomics over addresses)
constexpr int kStatSlots = 32;

}  // namespace mi
