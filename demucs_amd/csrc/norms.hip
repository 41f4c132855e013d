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
// Half modes: `img` (optional) receives the tensor the NEXT matrix product reads -- y for MODE 0 / 2, x for MODE 1 -- as that
// product's 16-bit operand image [C / 8][img_n][8] (column b T + t; gemm_half.hip): one coalesced 16-byte store per eight
// channels and token, so that the projection moves it global -> LDS by DMA instead of loading and converting float32.
template <int MODE>
__global__ __launch_bounds__(256) void token_tile_kernel(const float *__restrict__ x, int C, int T, const float *__restrict__ w,
                                                         const float *__restrict__ bvec, const float *__restrict__ pe,
                                                         const float2 *__restrict__ gstat, float eps, float *__restrict__ y,
                                                         float2 *__restrict__ ostat, uint4 *__restrict__ img, int64_t img_n, int img_dtype) {
    __shared__ float red[4][64][2];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: w[c], b[c] become scalar loads
    const int t = blockIdx.x * 64 + lane;
    const bool ok = t < T;
    const size_t base = (size_t)blockIdx.y * C * T + (ok ? t : 0);
    const int cw = C / 4, c0 = wv * cw;
    const float inv_c = 1.0f / (float)C;
    float mean = 0.f, rstd = 1.f;
    if (MODE == 0 || MODE == 1) {                     // statistics of the input over channels
        const float x0 = x[base];
        float s1 = 0.f, s2 = 0.f;
        if (MODE == 1 && img) {
            for (int c = c0; c < c0 + cw; c += 8) {
                float q[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    q[e] = x[base + (size_t)(c + e) * T];
                    const float v = q[e] - x0;
                    s1 += v; s2 += v * v;
                }
                if (ok) img[(size_t)(c >> 3) * img_n + (size_t)blockIdx.y * T + t] =
                    make_uint4(pack_half2(img_dtype, q[0], q[1]), pack_half2(img_dtype, q[2], q[3]), pack_half2(img_dtype, q[4], q[5]),
                               pack_half2(img_dtype, q[6], q[7]));
            }
        } else
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
    // eight channels per trip: their loads are issued together (one channel per trip left ONE load in flight per wave: the trip
    // waited for x, w[c] and b[c] before the next address was formed); same per-element arithmetic and summation order
    int c = c0;
    for (; c + 8 <= c0 + cw; c += 8) {
        float xv[8], pv[8], q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = x[base + (size_t)(c + e) * T];
        if (MODE == 0 && pe) {
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = pe[(size_t)(c + e) * T + (ok ? t : 0)];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (xv[e] - mean) * rstd * w[c + e] + bvec[c + e];
            if (MODE == 0 && pe) v += pv[e];
            if (ok) y[base + (size_t)(c + e) * T] = v;
            const float dv = v - y0;
            s1 += dv; s2 += dv * dv;
            q8[e] = v;
        }
        if (img && ok)                                    // cw is a multiple of 8 (checked by the launcher)
            img[(size_t)(c >> 3) * img_n + (size_t)blockIdx.y * T + t] =
                make_uint4(pack_half2(img_dtype, q8[0], q8[1]), pack_half2(img_dtype, q8[2], q8[3]), pack_half2(img_dtype, q8[4], q8[5]),
                           pack_half2(img_dtype, q8[6], q8[7]));
    }
    for (; c < c0 + cw; ++c) {                            // widths whose quarter is not a multiple of 8 (never with an image)
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

// ---- DConv (implicit-GEMM route): GroupNorm(1) + GELU of the hidden tensor AND the second GroupNorm's statistics ------
// The 1x1 conv that follows, z = W g + b over 2C channels, is normalised by GroupNorm(1, 2C) (demucs.py:141-142), whose
// statistics over a row's columns follow from  s = sum g  and  G = sum g g^T  (h x h, h = C / 8 or C / 4):
//     sum z = colsum(W) . s + n sum b,      sum z^2 = <W^T W, G> + 2 (W^T b) . s + n sum b^2.
// One pass instead of a statistics-only evaluation of the 2C x h GEMM: this kernel normalises 128 columns of a row in
// place, keeps them as an LDS tile Gs[HP][128 or 64] (row h = ones, so G[.][h] = s; rows above are zero) and forms the tile's
// Gram blocks on the float32 matrix pipe (v_mfma_f32_32x32x2_f32: exact products), upper block triangle only, added
// to the row's float64 accumulators.  grid (ceil(cols / column tile), rows); HP = h + 1 rounded up to 32.
typedef float gram_f32x16 __attribute__((ext_vector_type(16)));
template <int HP>
__global__ __launch_bounds__(256) void gn_gelu_gram_kernel(float *__restrict__ x, int h, int Cs, int D1, int D2, int pitch, int row_mode,
                                                           const float2 *__restrict__ st, const float *__restrict__ w,
                                                           const float *__restrict__ bvec, double *__restrict__ gram, int slots) {
    constexpr int CT = HP > 96 ? 64 : 128, CG = 256 / CT, LDG = CT + 1, NT = HP / 32;      // column tile: <= 48 KiB of LDS
    __shared__ float Gs[HP][LDG];
    const int tid = threadIdx.x, row = blockIdx.y, col = tid & (CT - 1), cg = tid / CT;
    const int b = row_mode ? row / D1 : row, d1 = row_mode ? row % D1 : 0;
    const int cols = row_mode ? D2 : D1 * D2;                    // valid columns of a statistics row (time branch: D1 == 1)
    const int q = blockIdx.x * CT + col;
    const bool ok = q < cols;
    const float2 s = st[row];
    float *xp = x + (size_t)b * Cs * D1 * pitch + (size_t)d1 * pitch + q;
    // all loads first: a load after a store into the same tensor is not hoisted above it, and the chain of round trips
    // would cost more than the statistics GEMM this pass replaces
    float v[HP / CG];
#pragma unroll
    for (int k = 0; k < HP / CG; ++k) {
        const int c = cg + k * CG;
        v[k] = (c < h && ok) ? xp[(size_t)c * D1 * pitch] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < HP / CG; ++k) {
        const int c = cg + k * CG;
        float g = 0.f;
        if (c < h && ok) {
            g = gelu_exact((v[k] - s.x) * s.y * w[c] + bvec[c]);
            xp[(size_t)c * D1 * pitch] = g;
        } else if (c == h && ok) {
            g = 1.f;
        }
        Gs[c][col] = g;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    double *gout = gram + ((size_t)row * slots + blockIdx.x % slots) * HP * HP;
    int t = 0;
    for (int ti = 0; ti < NT; ++ti)
        for (int tj = ti; tj < NT; ++tj, ++t) {
            if (t % 4 != wave) continue;
            if (ti * 32 > h) continue;                           // block rows above the ones row are zero
            gram_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *ga = &Gs[ti * 32 + li][lh], *gb = &Gs[tj * 32 + li][lh];
#pragma unroll 8
            for (int k = 0; k < CT; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[k], gb[k], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, j = tj * 32 + li;
                if (i <= h && j <= h) atomicAdd(gout + (size_t)i * HP + j, (double)acc[r]);
            }
        }
}

// statistics of z from the Gram accumulators of gn_gelu_gram_kernel (re-zeroed here: self-cleaning like the statistic slots).
// wt: HP x HP float64 weights of the accumulated entries (host: W^T W on the diagonal blocks, doubled above them, 2 W^T b in
// column h); ct: HP weights of column h for sum z (colsum W).  One workgroup per statistics row.
__global__ __launch_bounds__(256) void gram_finalize_kernel(double *__restrict__ gram, int HP, int slots, int h, const double *__restrict__ wt,
                                                            const double *__restrict__ ct, double sum_b, double sum_bsq, double cols,
                                                            double count, float eps, float2 *__restrict__ out) {
    __shared__ double red[2][4];
    const int row = blockIdx.x, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int e = tid; e < HP * HP; e += 256) {
        const int i = e / HP, j = e % HP;
        if (i > h || j > h || (j >> 5) < (i >> 5)) continue;     // never written
        double *p = gram + (size_t)row * slots * HP * HP + e;
        double v[8];                                             // all loads before the stores that re-zero the slots
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) v[sl] = sl < slots ? p[(size_t)sl * HP * HP] : 0.0;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
            if (sl < slots) p[(size_t)sl * HP * HP] = 0.0;
        const double acc = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        s2 += wt[e] * acc;
        if (j == h) s1 += ct[i] * acc;
    }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3] + cols * sum_b;
        s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3] + cols * sum_bsq;
        const double mean = s1 / count;
        double m2 = s2 - s1 * mean;
        if (m2 < 0.0) m2 = 0.0;
        out[row] = make_float2((float)mean, 1.0f / sqrtf((float)(m2 / count) + eps));
    }
}

int gram_hp(int h) { return (h + 1 + 31) / 32 * 32; }

int launch_gn_gelu_gram(float *x, int B, int h, int Cs, int D1, int D2, int pitch, int row_mode, const float2 *stats, const float *w,
                        const float *b, double *gram, int slots, hipStream_t st) {
    const int HP = gram_hp(h), rows = row_mode ? B * D1 : B, cols = row_mode ? D2 : D1 * D2;
    MI_REQUIRE(HP <= 128, "gn_gelu_gram: %d hidden channels not instantiated", h);
    MI_REQUIRE(row_mode || D1 == 1 || pitch == D2, "gn_gelu_gram: a per-item row needs contiguous positions");
    const dim3 grid(ceil_div(cols, HP > 96 ? 64 : 128), rows);
    if (HP == 32) hipLaunchKernelGGL(gn_gelu_gram_kernel<32>, grid, dim3(256), 0, st, x, h, Cs, D1, D2, pitch, row_mode, stats, w, b, gram, slots);
    else if (HP == 64) hipLaunchKernelGGL(gn_gelu_gram_kernel<64>, grid, dim3(256), 0, st, x, h, Cs, D1, D2, pitch, row_mode, stats, w, b, gram, slots);
    else if (HP == 96) hipLaunchKernelGGL(gn_gelu_gram_kernel<96>, grid, dim3(256), 0, st, x, h, Cs, D1, D2, pitch, row_mode, stats, w, b, gram, slots);
    else hipLaunchKernelGGL(gn_gelu_gram_kernel<128>, grid, dim3(256), 0, st, x, h, Cs, D1, D2, pitch, row_mode, stats, w, b, gram, slots);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_gram_finalize(double *gram, int rows, int h, int slots, const double *wt, const double *ct, double sum_b, double sum_bsq,
                         double cols, double count, float eps, float2 *out, hipStream_t st) {
    MI_REQUIRE(slots >= 1 && slots <= 8, "gram_finalize: %d accumulator slots (1..8)", slots);
    hipLaunchKernelGGL(gram_finalize_kernel, dim3(rows), dim3(256), 0, st, gram, gram_hp(h), slots, h, wt, ct, sum_b, sum_bsq, cols, count, eps, out);
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

static int check_img(int C, const void *img, int64_t img_n, int B, int T, int dtype) {
    MI_REQUIRE(!img || (C % 32 == 0 && img_n >= (int64_t)B * T && ((uintptr_t)img & 15) == 0 && (dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16)),
               "token kernel: operand image needs C %% 32 == 0, >= B * T columns, 16-byte alignment and a half dtype");
    return MI_OK;
}

int launch_layernorm_cf(const float *x, int B, int C, int T, const float *w, const float *b, const float *pe, float *y,
                        float2 *ostat, hipStream_t st, void *img, int64_t img_n, int img_dtype) {
    MI_REQUIRE(C % 4 == 0, "layernorm: C %% 4 != 0");
    MI_TRY(check_img(C, img, img_n, B, T, img_dtype));
    hipLaunchKernelGGL(token_tile_kernel<0>, dim3(ceil_div(T, 64), B), dim3(256), 0, st, x, C, T, w, b, pe, nullptr, 1e-5f, y, ostat,
                       (uint4 *)img, img_n, img_dtype);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_token_stats(const float *x, int B, int C, int T, float2 *ostat, hipStream_t st, void *img, int64_t img_n, int img_dtype) {
    MI_REQUIRE(C % 4 == 0, "token_stats: C %% 4 != 0");
    MI_TRY(check_img(C, img, img_n, B, T, img_dtype));
    hipLaunchKernelGGL(token_tile_kernel<1>, dim3(ceil_div(T, 64), B), dim3(256), 0, st, x, C, T, nullptr, nullptr, nullptr, nullptr,
                       1e-5f, nullptr, ostat, (uint4 *)img, img_n, img_dtype);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_gn_apply_tokstats(const float *x, int B, int C, int T, const float2 *gstat, const float *w, const float *b, float *y,
                             float2 *ostat, hipStream_t st, void *img, int64_t img_n, int img_dtype) {
    MI_REQUIRE(C % 4 == 0, "gn_apply: C %% 4 != 0");
    MI_TRY(check_img(C, img, img_n, B, T, img_dtype));
    hipLaunchKernelGGL(token_tile_kernel<2>, dim3(ceil_div(T, 64), B), dim3(256), 0, st, x, C, T, w, b, nullptr, gstat, 1e-5f, y, ostat,
                       (uint4 *)img, img_n, img_dtype);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
