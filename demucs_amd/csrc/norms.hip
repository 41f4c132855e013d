// Normalisation kernels (HBM-bound, wave64 reductions): per-item input normalisation
// (reference: demucs/htdemucs.py:545-554), LayerNorm over channels of channel-first tokens,
// GroupNorm(1) over (tokens, channels) (demucs/transformer.py:258-268), statistics finalisation.
#include "common.h"
#include "kernels.h"

namespace mi {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-wide (256 threads) sum of two doubles -> thread 0
__device__ __forceinline__ void block_sum2(double &a, double &b, double *red /*[8]*/) {
    a = wave_sum(a); b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) { a = red[0] + red[2] + red[4] + red[6]; b = red[1] + red[3] + red[5] + red[7]; }
}

// sum / sum-of-squares of `count` contiguous floats per row -> stats[row][slot][2] (fp64 atomics).
// grid (nblk, rows)
__global__ __launch_bounds__(256) void row_stats_kernel(const float *__restrict__ x, int64_t count, int64_t row_stride,
                                                        double *__restrict__ stats) {
    __shared__ double red[8];
    const float *p = x + (size_t)blockIdx.y * row_stride;
    const int64_t per = (count + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(count, lo + per);
    double s1 = 0.0, s2 = 0.0;
    for (int64_t base = lo; base < hi; base += 256 * 16) {      // fp32 partials over <=16 elements, fp64 across
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t i = base + threadIdx.x + 256 * j;
            if (i < hi) { const float v = p[i]; a += v; q += v * v; }
        }
        s1 += a; s2 += q;
    }
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) {
        double *dst = stats + ((size_t)blockIdx.y * kStatSlots + (blockIdx.x % kStatSlots)) * 2;
        atomicAdd(dst, s1); atomicAdd(dst + 1, s2);
    }
}

// mode 0: GroupNorm   -> out_a[row] = (mean, 1/sqrt(var_biased + eps))
// mode 1: item norm   -> out_a[row] = (mean, 1/(eps + std_unbiased)), out_b[row] = (mean, std_unbiased)
// The slots are zeroed again after being read: the workspace statistics buffers are zero at creation and
// every producer pass is followed by exactly one finalize, so no memset launch is needed between uses.
__global__ void finalize_stats_kernel(double *__restrict__ stats, int rows, double count, float eps, int mode,
                                      float2 *__restrict__ out_a, float2 *__restrict__ out_b) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < kStatSlots; ++s) {
        double *p = stats + ((size_t)row * kStatSlots + s) * 2;
        s1 += p[0]; s2 += p[1];
        p[0] = 0.0; p[1] = 0.0;
    }
    const double mean = s1 / count;
    double m2 = s2 - s1 * mean;          // sum (x - mean)^2
    if (m2 < 0.0) m2 = 0.0;
    if (mode == 0) {
        const float var = (float)(m2 / count);
        out_a[row] = make_float2((float)mean, 1.0f / sqrtf(var + eps));
    } else {
        const float sd = (float)sqrt(m2 / (count - 1.0));
        out_a[row] = make_float2((float)mean, 1.0f / (eps + sd));
        if (out_b) out_b[row] = make_float2((float)mean, sd);
    }
}

// y[row][i] = (x[row][i] - mean[row]) * inv[row]
__global__ __launch_bounds__(256) void row_affine_kernel(const float *__restrict__ x, int64_t count, const float2 *__restrict__ norm,
                                                         float *__restrict__ y) {
    const float2 nm = norm[blockIdx.y];
    const size_t base = (size_t)blockIdx.y * count;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        y[base + i] = (x[base + i] - nm.x) * nm.y;
}

// Token-tile kernel on channel-first tokens x[b][C][T]: grid (ceil(T/64), B), block 256; lane = token, wave w
// owns channels [w*C/4, (w+1)*C/4).  One template, three uses:
//   MODE 0  LayerNorm (+ optional additive table pe[C][T]) -> y, and (mean, rstd) of y over channels -> ostat
//   MODE 1  per-token (mean, rstd) of x only -> ostat                      (LayerNorm folded into the next GEMM)
//   MODE 2  GroupNorm(1) apply y = (x - gm[b]) * gr[b] * w[c] + b[c] -> y, and (mean, rstd) of y -> ostat
// Variances use sums shifted by the token's first value (no cancellation), float32.
template <int MODE>
__global__ __launch_bounds__(256) void token_tile_kernel(const float *__restrict__ x, int C, int T, const float *__restrict__ w,
                                                         const float *__restrict__ bvec, const float *__restrict__ pe,
                                                         const float2 *__restrict__ gstat, float eps, float *__restrict__ y,
                                                         float2 *__restrict__ ostat) {
    __shared__ float red[4][64][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const bool ok = t < T;
    const size_t base = (size_t)blockIdx.y * C * T + (ok ? t : 0);
    const int cw = C / 4, c0 = wv * cw;
    const float inv_c = 1.0f / (float)C;
    float mean = 0.f, rstd = 1.f;
    if (MODE == 0 || MODE == 1) {                     // statistics of the input over channels
        const float x0 = x[base];
        float s1 = 0.f, s2 = 0.f;
        for (int c = c0; c < c0 + cw; ++c) {
            const float v = x[base + (size_t)c * T] - x0;
            s1 += v; s2 += v * v;
        }
        red[wv][lane][0] = s1; red[wv][lane][1] = s2;
        __syncthreads();
        s1 = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
        s2 = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
        const float dm = s1 * inv_c;
        mean = x0 + dm;
        rstd = 1.0f / sqrtf(fmaxf(s2 * inv_c - dm * dm, 0.f) + eps);
        if (MODE == 1) {
            if (ok && wv == 0) ostat[(size_t)blockIdx.y * T + t] = make_float2(mean, rstd);
            return;
        }
        __syncthreads();                              // red[] is reused below
    } else {
        const float2 g = gstat[blockIdx.y];
        mean = g.x; rstd = g.y;
    }
    // produce y and the statistics of y (shifted by y at channel 0 of this token)
    float y0;
    {
        float v = (x[base] - mean) * rstd * w[0] + bvec[0];
        if (MODE == 0 && pe) v += pe[ok ? t : 0];
        y0 = v;
    }
    float s1 = 0.f, s2 = 0.f;
    for (int c = c0; c < c0 + cw; ++c) {
        float v = (x[base + (size_t)c * T] - mean) * rstd * w[c] + bvec[c];
        if (MODE == 0 && pe) v += pe[(size_t)c * T + (ok ? t : 0)];
        if (ok) y[base + (size_t)c * T] = v;
        const float dv = v - y0;
        s1 += dv; s2 += dv * dv;
    }
    if (!ostat) return;
    red[wv][lane][0] = s1; red[wv][lane][1] = s2;
    __syncthreads();
    if (ok && wv == 0) {
        s1 = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
        s2 = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
        const float dm = s1 * inv_c;
        ostat[(size_t)blockIdx.y * T + t] = make_float2(y0 + dm, 1.0f / sqrtf(fmaxf(s2 * inv_c - dm * dm, 0.f) + eps));
    }
}

// x[b][c][d1][d2] <- gelu((x - mean[row]) * rstd[row] * w[c] + bias[c]) in place; row = b*D1 + d1 (row_mode 1) or b.
// GroupNorm(1) + GELU of the DConv hidden tensor (demucs.py:139); Cs = channels allocated per item.  grid (ceil(D1*D2/256), C, B)
__global__ __launch_bounds__(256) void gn_gelu_kernel(float *__restrict__ x, int C, int Cs, int D1, int D2, int row_mode,
                                                      const float2 *__restrict__ st, const float *__restrict__ w,
                                                      const float *__restrict__ bvec) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= D1 * D2) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const float2 s = st[row_mode ? b * D1 + p / D2 : b];
    const size_t i = ((size_t)b * Cs + c) * D1 * D2 + p;
    x[i] = gelu_exact((x[i] - s.x) * s.y * w[c] + bvec[c]);
}

int launch_gn_gelu(float *x, int B, int C, int Cs, int D1, int D2, int row_mode, const float2 *stats, const float *w, const float *b,
                   hipStream_t st) {
    hipLaunchKernelGGL(gn_gelu_kernel, dim3(ceil_div((int64_t)D1 * D2, 256), C, B), dim3(256), 0, st, x, C, Cs, D1, D2, row_mode, stats, w, b);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_row_stats(const float *x, int rows, int64_t count, int64_t row_stride, double *stats, hipStream_t st) {
    const int nblk = (int)std::min<int64_t>(256, (count + 4095) / 4096);
    hipLaunchKernelGGL(row_stats_kernel, dim3(nblk, rows), dim3(256), 0, st, x, count, row_stride, stats);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_finalize_stats(double *stats, int rows, double count, float eps, int mode, float2 *out_a, float2 *out_b,
                          hipStream_t st) {
    hipLaunchKernelGGL(finalize_stats_kernel, dim3(ceil_div(rows, 64)), dim3(64), 0, st, stats, rows, count, eps, mode, out_a, out_b);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// y[row][i] = x[row][i] * std[row] + mean[row]   (htdemucs.py:626,656), denorm = (mean, std)
__global__ __launch_bounds__(256) void row_denorm_kernel(const float *__restrict__ x, int64_t count, const float2 *__restrict__ denorm,
                                                         float *__restrict__ y) {
    const float2 d = denorm[blockIdx.y];
    const size_t base = (size_t)blockIdx.y * count;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        y[base + i] = x[base + i] * d.y + d.x;
}

int launch_row_denorm(const float *x, int rows, int64_t count, const float2 *denorm, float *y, hipStream_t st) {
    const int nblk = (int)std::min<int64_t>(2048, (count + 255) / 256);
    hipLaunchKernelGGL(row_denorm_kernel, dim3(nblk, rows), dim3(256), 0, st, x, count, denorm, y);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_row_affine(const float *x, int rows, int64_t count, const float2 *norm, float *y, hipStream_t st) {
    const int nblk = (int)std::min<int64_t>(1024, (count + 255) / 256);
    hipLaunchKernelGGL(row_affine_kernel, dim3(nblk, rows), dim3(256), 0, st, x, count, norm, y);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_layernorm_cf(const float *x, int B, int C, int T, const float *w, const float *b, const float *pe, float *y,
                        float2 *ostat, hipStream_t st) {
    MI_REQUIRE(C % 4 == 0, "layernorm: C %% 4 != 0");
    hipLaunchKernelGGL(token_tile_kernel<0>, dim3(ceil_div(T, 64), B), dim3(256), 0, st, x, C, T, w, b, pe, nullptr, 1e-5f, y, ostat);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_token_stats(const float *x, int B, int C, int T, float2 *ostat, hipStream_t st) {
    MI_REQUIRE(C % 4 == 0, "token_stats: C %% 4 != 0");
    hipLaunchKernelGGL(token_tile_kernel<1>, dim3(ceil_div(T, 64), B), dim3(256), 0, st, x, C, T, nullptr, nullptr, nullptr, nullptr,
                       1e-5f, nullptr, ostat);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_gn_apply_tokstats(const float *x, int B, int C, int T, const float2 *gstat, const float *w, const float *b, float *y,
                             float2 *ostat, hipStream_t st) {
    MI_REQUIRE(C % 4 == 0, "gn_apply: C %% 4 != 0");
    hipLaunchKernelGGL(token_tile_kernel<2>, dim3(ceil_div(T, 64), B), dim3(256), 0, st, x, C, T, w, b, nullptr, gstat, 1e-5f, y, ostat);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
