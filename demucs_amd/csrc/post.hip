// Track-level elementwise work around `apply_model` on the device (all HBM-bound, one pass each):
//   * `Separator.separate_tensor`'s normalisation by the mono mean / std and its inverse (reference: demucs/api.py:267-290);
//   * `prevent_clip` (demucs/audio.py:218-234) and the `--two-stems` sums (demucs/separate.py:189-218) on the stems.
// Scalars (mean, std, peak) stay in device memory between the kernels: nothing here synchronises with the host.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace mi {

constexpr int kPostBlocks = 1024;      // partial-sum slots of the statistics pass (fixed: the result does not depend on the grid)

// ref = wav.mean(0) (float32, channel sum then / channels, api.py:267); partial sums of ref and ref^2 in float64.
// grid kPostBlocks x 256; block b reduces the positions p = b*256 + t, + kPostBlocks*256, ...: a fixed assignment.
__global__ __launch_bounds__(256) void mono_stats_partial_kernel(const float *__restrict__ wav, int channels, int64_t length,
                                                                 double *__restrict__ part) {
    double s = 0.0, q = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < length; p += (int64_t)kPostBlocks * 256) {
        float a = wav[p];
        for (int c = 1; c < channels; ++c) a = __fadd_rn(a, wav[(size_t)c * length + p]);
        const float m = __fdiv_rn(a, (float)channels);
        s += (double)m;
        q += (double)m * (double)m;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { sh[0][threadIdx.x] += sh[0][threadIdx.x + w]; sh[1][threadIdx.x] += sh[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0][0]; part[2 * blockIdx.x + 1] = sh[1][0]; }
}

// stats[0] = ref.mean(), stats[1] = ref.std() + 1e-8 (unbiased, torch's default), both float32.  one block of 256
__global__ __launch_bounds__(256) void mono_stats_final_kernel(const double *__restrict__ part, int64_t length, float *__restrict__ stats) {
    __shared__ double sh[2][256];
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < kPostBlocks; i += 256) { s += part[2 * i]; q += part[2 * i + 1]; }
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { sh[0][threadIdx.x] += sh[0][threadIdx.x + w]; sh[1][threadIdx.x] += sh[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)length, mean = sh[0][0] / n;
        const double var = length > 1 ? fmax(sh[1][0] - n * mean * mean, 0.0) / (n - 1.0) : NAN;     // torch: std of one element is nan
        stats[0] = (float)mean;
        stats[1] = __fadd_rn((float)sqrt(var), 1e-8f);
    }
}

// mode 0: x = (x - mean) / s  (api.py:268-269: two separately rounded float32 ops); mode 1: x = x * s + mean (api.py:285-288)
__global__ __launch_bounds__(256) void track_affine_kernel(float *__restrict__ x, int64_t n, const float *__restrict__ stats, int mode) {
    const float mean = stats[0], s = stats[1];
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n && (((uintptr_t)x) & 15) == 0) {
        float4 v = *reinterpret_cast<float4 *>(x + i4);
        if (mode == 0) {
            v.x = __fdiv_rn(__fsub_rn(v.x, mean), s); v.y = __fdiv_rn(__fsub_rn(v.y, mean), s);
            v.z = __fdiv_rn(__fsub_rn(v.z, mean), s); v.w = __fdiv_rn(__fsub_rn(v.w, mean), s);
        } else {
            v.x = __fadd_rn(__fmul_rn(v.x, s), mean); v.y = __fadd_rn(__fmul_rn(v.y, s), mean);
            v.z = __fadd_rn(__fmul_rn(v.z, s), mean); v.w = __fadd_rn(__fmul_rn(v.w, s), mean);
        }
        *reinterpret_cast<float4 *>(x + i4) = v;
    } else {
        for (int64_t i = i4; i < n && i < i4 + 4; ++i)
            x[i] = mode == 0 ? __fdiv_rn(__fsub_rn(x[i], mean), s) : __fadd_rn(__fmul_rn(x[i], s), mean);
    }
}

// peak = max |x| as the bit pattern of a non-negative float (unsigned order == float order; max is order-independent).  The
// reduction runs on the bit patterns: |NaN| orders above +inf, so a NaN sample reaches `peak` as torch's abs().max() propagates it
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ peak) {
    unsigned m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = max(m, __float_as_uint(fabsf(x[i])));
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(peak, m);
}

// mode 1 "rescale": x / max(1.01 * peak, 1); 2 "clamp": clamp(x, -0.99, 0.99); 3 "tanh"   (audio.py:225-231)
__global__ __launch_bounds__(256) void prevent_clip_kernel(const float *__restrict__ x, int64_t n, int mode, const unsigned *__restrict__ peak,
                                                           float *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    if (mode == 1) {
        const float d = __fmul_rn(__uint_as_float(*peak), 1.01f);
        r = (d > 1.0f || d != d) ? __fdiv_rn(v, d) : v;       // a NaN peak divides everything (python's max(nan_tensor, 1) keeps the NaN)
    } else if (mode == 2) {
        r = fminf(fmaxf(v, -0.99f), 0.99f);
    } else {
        r = tanhf(v);
    }
    y[i] = r;
}

struct StemPtrs { const float *p[8]; };

// mode 0 "add": y = 0 + stem_a + stem_b + ... over every stem but `sel`, in index order (separate.py:205-208);
// mode 1 "minus": y = origin - stem_sel (separate.py:197)
__global__ __launch_bounds__(256) void two_stems_kernel(StemPtrs stems, int S, int sel, const float *__restrict__ origin, int mode, int64_t n,
                                                        float *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (mode == 1) { y[i] = __fsub_rn(origin[i], stems.p[sel][i]); return; }
    float a = 0.f;
    for (int k = 0; k < S; ++k)
        if (k != sel) a = __fadd_rn(a, stems.p[k][i]);
    y[i] = a;
}

int post_stats_scratch_bytes() { return kPostBlocks * 2 * (int)sizeof(double); }

int launch_mono_stats(const float *wav, int channels, int64_t length, double *scratch, float *stats, hipStream_t st) {
    hipLaunchKernelGGL(mono_stats_partial_kernel, dim3(kPostBlocks), dim3(256), 0, st, wav, channels, length, scratch);
    MI_CHECK_LAUNCH();
    hipLaunchKernelGGL(mono_stats_final_kernel, dim3(1), dim3(256), 0, st, scratch, length, stats);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_track_affine(float *x, int64_t n, const float *stats, int mode, hipStream_t st) {
    hipLaunchKernelGGL(track_affine_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, st, x, n, stats, mode);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_prevent_clip(const float *x, int64_t n, int mode, unsigned *peak, float *y, hipStream_t st) {
    if (mode == 1) {
        MI_HIP(hipMemsetAsync(peak, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL(absmax_kernel, dim3((int)std::min<int64_t>(2048, ceil_div(n, 256))), dim3(256), 0, st, x, n, peak);
        MI_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(prevent_clip_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, x, n, mode, peak, y);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_two_stems(const float *const *stems, int S, int sel, const float *origin, int mode, int64_t n, float *y, hipStream_t st) {
    StemPtrs sp{};
    for (int k = 0; k < S; ++k) sp.p[k] = stems[k];
    hipLaunchKernelGGL(two_stems_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, sp, S, sel, origin, mode, n, y);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
