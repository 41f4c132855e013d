// Fused DConv for the TIME branch (one GroupNorm row per batch item: [C][L] with L up to 85 995), C = 48 / 96
// (reference: demucs/demucs.py:133-154 as used by HEncLayer / HDecLayer with freq=False, demucs/hdemucs.py:145,316):
//     dilated conv3 C->C/8  ->  GroupNorm(1) -> GELU  ->  1x1 C/8->2C  ->  GroupNorm(1)  ->  GLU -> LayerScale -> +x
// The implicit-GEMM route pads M = C/8 = 6 / 12 hidden channels to a 32-row MFMA tile (19 % / 37 % useful), evaluates
// the 1x1 twice on the matrix cores and runs the GroupNorm+GELU as a pass of its own.  Here a lane owns 6 adjacent
// columns and keeps the (C/8 x 6) hidden values in registers (fp32 VALU FMAs, weights broadcast from LDS: every FLOP is
// useful), in three streaming passes per residual layer, because GroupNorm(1) needs statistics over the WHOLE row:
//   A  conv3 on a column tile (+ dilation halo)  -> h (C/8 x L, 1/6 of x) + sum / sum^2 of h           [reads x]
//   B  g = GELU(GN(h)) recomputed on the fly      -> sum_t g and the Gram matrix sum_t g g^T            [reads h]
//   C  g again, z = W3 g + b3, GN, GLU, LayerScale, + x -> out                                          [reads h, x]
// The second GroupNorm's statistics never need z itself: with s = sum_t g[t] and G = sum_t g[t] g[t]^T,
//     sum z   = colsum(W3) . s + L sum(b3),      sum z^2 = <W3^T W3, G> + 2 (W3^T b3) . s + L |b3|^2,
// so pass B costs H (H + 3) / 2 products per column instead of the 2C x H of a second 1x1 evaluation.
#include "common.h"
#include "kernels.h"

namespace mi {

constexpr int kTC = 6;                   // columns per lane
constexpr int kGramMax = 96;             // doubles per Gram slot: H (H + 1) / 2 + H <= 90 for H = 12

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// columns [t0 - 2, t0 + 8) of one channel row as five float2, zero outside [0, Lv)
struct Taps { float v[10]; };
__device__ __forceinline__ Taps load_taps(const float *row, int t0, int Lv, int Lp) {
    Taps r;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int p = t0 - 2 + 2 * q;                    // even
        float2 f = make_float2(0.f, 0.f);
        if (p >= 0 && p < Lp) f = *reinterpret_cast<const float2 *>(row + p);
        r.v[2 * q] = (p >= 0 && p < Lv) ? f.x : 0.f;
        r.v[2 * q + 1] = (p + 1 >= 0 && p + 1 < Lv) ? f.y : 0.f;
    }
    return r;
}

// ---- pass A ------------------------------------------------------------------------------------------------------
template <int C, int H, int DIL>
__global__ __launch_bounds__(256) void dconv_t_conv3_kernel(const DConvRowLayer L, const float *__restrict__ x, float *__restrict__ hbuf,
                                                            int Lv, int Lp, double *__restrict__ stats) {
    constexpr int HA = (H + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float w0s[C * 3 * HA];
    __shared__ float b0s[HA];
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int i = tid; i < C * 3 * HA; i += 256) w0s[i] = L.w0[i];
    if (tid < HA) b0s[tid] = L.b0[tid];
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + tid) * kTC;
    const bool on = t0 < Lv;
    const float *xb = x + (size_t)b * C * Lp;
    v2f hid[kTC][HA / 2];        // pairs of hidden channels: one v_pk_fma_f32 per tap and pair
#pragma unroll
    for (int j = 0; j < kTC; ++j)
#pragma unroll
        for (int m = 0; m < HA / 2; ++m) hid[j][m] = (v2f){b0s[2 * m], b0s[2 * m + 1]};
    if (on) {
        Taps nx = load_taps(xb, t0, Lv, Lp);
#pragma unroll 2
        for (int c = 0; c < C; ++c) {
            v2f xs[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) xs[i] = splat2(nx.v[i]);
            if (c + 1 < C) nx = load_taps(xb + (size_t)(c + 1) * Lp, t0, Lv, Lp);
            const float4 *wv = reinterpret_cast<const float4 *>(w0s + c * 3 * HA);
#pragma unroll
            for (int q = 0; q < HA / 4; ++q) {
                const float4 wa = wv[q], wb = wv[HA / 4 + q], wc = wv[2 * (HA / 4) + q];
#pragma unroll
                for (int j = 0; j < kTC; ++j) {
                    const v2f x0 = xs[2 + j - DIL], x1 = xs[2 + j], x2 = xs[2 + j + DIL];
                    hid[j][2 * q] = fma2((v2f){wc.x, wc.y}, x2, fma2((v2f){wb.x, wb.y}, x1, fma2((v2f){wa.x, wa.y}, x0, hid[j][2 * q])));
                    hid[j][2 * q + 1] = fma2((v2f){wc.z, wc.w}, x2, fma2((v2f){wb.z, wb.w}, x1, fma2((v2f){wa.z, wa.w}, x0, hid[j][2 * q + 1])));
                }
            }
        }
    }
    float p1 = 0.f, p2 = 0.f;
    if (on) {
        float *hb = hbuf + (size_t)b * HA * Lp + t0;
#pragma unroll
        for (int m = 0; m < H; ++m) {
#pragma unroll
            for (int j = 0; j < kTC; ++j)
                if (t0 + j < Lv) { const float v = hid[j][m >> 1][m & 1]; p1 += v; p2 += v * v; }
#pragma unroll
            for (int j = 0; j < kTC; j += 2)
                if (t0 + j < Lp) *reinterpret_cast<float2 *>(hb + (size_t)m * Lp + j) = make_float2(hid[j][m >> 1][m & 1], hid[j + 1][m >> 1][m & 1]);
        }
    }
    double s1 = wsum((double)p1), s2 = wsum((double)p2);
    __shared__ double red[8];
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s1; red[(tid >> 6) * 2 + 1] = s2; }
    __syncthreads();
    if (tid == 0) {
        double *dst = stats + ((size_t)b * kStatSlots + (blockIdx.x % kStatSlots)) * 2;
        atomicAdd(dst, red[0] + red[2] + red[4] + red[6]);
        atomicAdd(dst + 1, red[1] + red[3] + red[5] + red[7]);
    }
}

// h -> g = GELU((h - mean) rstd w + b) for the H hidden channels of this lane's 6 columns (0 outside [0, Lv))
template <int H, int HA>
__device__ __forceinline__ void load_g(const float *hb, int Lp, int t0, int Lv, float mu, float rs, const float *gw, const float *gb,
                                       float (&g)[kTC][HA]) {
#pragma unroll
    for (int m = 0; m < HA; ++m) {
        if (m < H) {
            const float a = rs * gw[m], c = gb[m] - mu * a;
#pragma unroll
            for (int j = 0; j < kTC; j += 2) {
                float2 f = make_float2(0.f, 0.f);
                if (t0 + j < Lp) f = *reinterpret_cast<const float2 *>(hb + (size_t)m * Lp + j);
                g[j][m] = t0 + j < Lv ? gelu_exact(fmaf(f.x, a, c)) : 0.f;
                g[j + 1][m] = t0 + j + 1 < Lv ? gelu_exact(fmaf(f.y, a, c)) : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kTC; ++j) g[j][m] = 0.f;
        }
    }
}

// ---- pass B ------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void dconv_t_gram_kernel(const DConvRowLayer L, const float *__restrict__ hbuf, int Lv, int Lp,
                                                           const float2 *__restrict__ st1, double *__restrict__ gram) {
    constexpr int HA = (H + 3) / 4 * 4, NG = H * (H + 1) / 2 + H;
    __shared__ double red[4][NG];
    __shared__ float gws[HA], gbs[HA];
    const int tid = threadIdx.x, b = blockIdx.y, lane = tid & 63, wv = tid >> 6;
    if (tid < HA) { gws[tid] = L.g1w[tid]; gbs[tid] = L.g1b[tid]; }
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + tid) * kTC;
    const float2 st = st1[b];
    float g[kTC][HA];
    if (t0 < Lv) load_g<H, HA>(hbuf + (size_t)b * HA * Lp + t0, Lp, t0, Lv, st.x, st.y, gws, gbs, g);
    else {
#pragma unroll
        for (int j = 0; j < kTC; ++j)
#pragma unroll
            for (int m = 0; m < HA; ++m) g[j][m] = 0.f;
    }
    int idx = 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
#pragma unroll
        for (int k = i; k < H; ++k) {
            float p = 0.f;
#pragma unroll
            for (int j = 0; j < kTC; ++j) p = fmaf(g[j][i], g[j][k], p);
            const double r = wsum((double)p);
            if (lane == 0) red[wv][idx] = r;
            ++idx;
        }
    }
#pragma unroll
    for (int i = 0; i < H; ++i) {
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < kTC; ++j) p += g[j][i];
        const double r = wsum((double)p);
        if (lane == 0) red[wv][idx] = r;
        ++idx;
    }
    __syncthreads();
    if (tid < NG)
        atomicAdd(gram + ((size_t)b * kStatSlots + (blockIdx.x % kStatSlots)) * kGramMax + tid,
                  red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

// statistics of z = W3 g + b3 over (2C, Lv) from the Gram sums; slots are zeroed again (self-cleaning like norms.hip).
// One wave per batch item: lane i owns Gram entry i (and i + 64), sums its 32 slots, weighs it, one butterfly.
__global__ __launch_bounds__(64) void dconv_t_gram_finalize_kernel(double *__restrict__ gram, int H, const double *__restrict__ ga /*[H(H+1)/2]: A_ii, 2 A_ik*/,
                                                                   const double *__restrict__ gv /*[H]: 2 W3^T b3*/,
                                                                   const double *__restrict__ gc /*[H]: colsum W3*/, double sum_b3,
                                                                   double sum_b3sq, double cols, double count, float eps,
                                                                   float2 *__restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nq = H * (H + 1) / 2;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < nq + H; i += 64) {
        double acc = 0.0;
        for (int s = 0; s < kStatSlots; ++s) {
            double *p = gram + ((size_t)b * kStatSlots + s) * kGramMax + i;
            acc += *p;
            *p = 0.0;
        }
        if (i < nq) s2 += ga[i] * acc;
        else { s2 += gv[i - nq] * acc; s1 += gc[i - nq] * acc; }
    }
    s1 = wsum(s1) + cols * sum_b3;
    s2 = wsum(s2) + cols * sum_b3sq;
    if (lane == 0) {
        const double mean = s1 / count;
        double m2 = s2 - s1 * mean;
        if (m2 < 0.0) m2 = 0.0;
        out[b] = make_float2((float)mean, 1.0f / sqrtf((float)(m2 / count) + eps));
    }
}

// ---- pass C ------------------------------------------------------------------------------------------------------
template <int C, int H>
__global__ __launch_bounds__(256) void dconv_t_out_kernel(const DConvRowLayer L, const float *__restrict__ x, const float *__restrict__ hbuf,
                                                          float *__restrict__ y, int Lv, int Lp, const float2 *__restrict__ st1,
                                                          const float2 *__restrict__ st2) {
    constexpr int HA = (H + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float w3s[4 * C * HA];     // [c][k] = (value w, value w, gate w, gate w): pre-splatted operands
    __shared__ float b3s[2 * C], g2ws[2 * C], g2bs[2 * C], lss[C], gws[HA], gbs[HA];
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int i = tid; i < 4 * C * HA; i += 256) {
        const int half = (i >> 1) & 1, k = (i >> 2) % HA, c = (i >> 2) / HA;
        w3s[i] = L.w3[(size_t)(c + half * C) * HA + k];
    }
    for (int i = tid; i < 2 * C; i += 256) { b3s[i] = L.b3[i]; g2ws[i] = L.g2w[i]; g2bs[i] = L.g2b[i]; }
    for (int i = tid; i < C; i += 256) lss[i] = L.ls[i];
    if (tid < HA) { gws[tid] = L.g1w[tid]; gbs[tid] = L.g1b[tid]; }
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + tid) * kTC;
    if (t0 >= Lv) return;
    const float2 s1 = st1[b], s2 = st2[b];
    float g[kTC][HA];
    load_g<H, HA>(hbuf + (size_t)b * HA * Lp + t0, Lp, t0, Lv, s1.x, s1.y, gws, gbs, g);
    const float *xb = x + (size_t)b * C * Lp + t0;
    float *yb = y + (size_t)b * C * Lp + t0;
    const float mu2 = s2.x, rs2 = s2.y;
    auto ld3 = [&](const float *p, float (&r)[kTC]) {
#pragma unroll
        for (int j = 0; j < kTC; j += 2) {
            float2 f = make_float2(0.f, 0.f);
            if (t0 + j < Lp) f = *reinterpret_cast<const float2 *>(p + j);
            r[j] = f.x; r[j + 1] = f.y;
        }
    };
    v2f gp[kTC / 2][HA];         // adjacent columns as packed pairs
#pragma unroll
    for (int jp = 0; jp < kTC / 2; ++jp)
#pragma unroll
        for (int m = 0; m < HA; ++m) gp[jp][m] = (v2f){g[2 * jp][m], g[2 * jp + 1][m]};
    float rn[kTC];
    ld3(xb, rn);
#pragma unroll 2
    for (int c = 0; c < C; ++c) {
        float r[kTC];
#pragma unroll
        for (int j = 0; j < kTC; ++j) r[j] = rn[j];
        if (c + 1 < C) ld3(xb + (size_t)(c + 1) * Lp, rn);
        const float4 *wp = reinterpret_cast<const float4 *>(w3s + (size_t)c * HA * 4);
        v2f zv[kTC / 2], zg[kTC / 2];
#pragma unroll
        for (int jp = 0; jp < kTC / 2; ++jp) { zv[jp] = splat2(b3s[c]); zg[jp] = splat2(b3s[c + C]); }
#pragma unroll
        for (int k = 0; k < HA; ++k) {
            const float4 w = wp[k];                      // (value w, value w, gate w, gate w) of hidden channel k
#pragma unroll
            for (int jp = 0; jp < kTC / 2; ++jp) {
                zv[jp] = fma2((v2f){w.x, w.y}, gp[jp][k], zv[jp]);
                zg[jp] = fma2((v2f){w.z, w.w}, gp[jp][k], zg[jp]);
            }
        }
        const float aA = rs2 * g2ws[c], aB = g2bs[c] - mu2 * aA, gA = rs2 * g2ws[c + C], gB = g2bs[c + C] - mu2 * gA, sc = lss[c];
        float o[kTC];
#pragma unroll
        for (int j = 0; j < kTC; ++j) o[j] = r[j] + sc * (fmaf(zv[j >> 1][j & 1], aA, aB) * sigmoid_f(fmaf(zg[j >> 1][j & 1], gA, gB)));
#pragma unroll
        for (int j = 0; j < kTC; j += 2)
            if (t0 + j < Lp) *reinterpret_cast<float2 *>(yb + (size_t)c * Lp + j) = make_float2(o[j], o[j + 1]);
    }
}

bool dconv_time_supported(int C, int Lp) { return (C == 48 || C == 96) && Lp % 2 == 0; }

// one residual layer: x (B, C, Lp) -> y (same shape, different buffer); hbuf holds (B, HA, Lp); stats / gram are the
// self-cleaning fp64 slot buffers ([B][kStatSlots][2] and [B][kStatSlots][kGramMax])
int launch_dconv_time_layer(const DConvTimeLayer &l, int C, int dil, int B, int Lv, int Lp, const float *x, float *y, float *hbuf,
                            double *stats, double *gram, float2 *st1, float2 *st2, hipStream_t st) {
    MI_REQUIRE(dconv_time_supported(C, Lp) && (dil == 1 || dil == 2), "dconv_time: unsupported C=%d Lp=%d dil=%d", C, Lp, dil);
    MI_REQUIRE(x != y && ((uintptr_t)x & 7) == 0 && ((uintptr_t)y & 7) == 0 && ((uintptr_t)hbuf & 7) == 0, "dconv_time: bad buffers");
    const int H = C / 8;
    const dim3 grid(ceil_div(Lv, 256 * kTC), B), blk(256);
#define MI_DT(CC, HH)                                                                                                         \
    do {                                                                                                                      \
        if (dil == 1) hipLaunchKernelGGL((dconv_t_conv3_kernel<CC, HH, 1>), grid, blk, 0, st, l.w, x, hbuf, Lv, Lp, stats);      \
        else hipLaunchKernelGGL((dconv_t_conv3_kernel<CC, HH, 2>), grid, blk, 0, st, l.w, x, hbuf, Lv, Lp, stats);               \
        MI_CHECK_LAUNCH();                                                                                                    \
        MI_TRY(launch_finalize_stats(stats, B, (double)HH * Lv, 1e-5f, 0, st1, nullptr, st));                                  \
        hipLaunchKernelGGL((dconv_t_gram_kernel<HH>), grid, blk, 0, st, l.w, hbuf, Lv, Lp, st1, gram);                           \
        MI_CHECK_LAUNCH();                                                                                                    \
        hipLaunchKernelGGL(dconv_t_gram_finalize_kernel, dim3(B), dim3(64), 0, st, gram, HH, l.gram_a, l.gram_v,               \
                           l.gram_c, l.sum_b3, l.sum_b3sq, (double)Lv, 2.0 * CC * Lv, 1e-5f, st2);                             \
        MI_CHECK_LAUNCH();                                                                                                    \
        hipLaunchKernelGGL((dconv_t_out_kernel<CC, HH>), grid, blk, 0, st, l.w, x, hbuf, y, Lv, Lp, st1, st2);                   \
        MI_CHECK_LAUNCH();                                                                                                    \
    } while (0)
    if (C == 48) MI_DT(48, 6);
    else MI_DT(96, 12);
#undef MI_DT
    (void)H;
    return MI_OK;
}

}  // namespace mi
