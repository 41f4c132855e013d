// Fused DConv for the frequency branch: one workgroup owns one (batch, frequency-bin) row [C][T] and runs
// BOTH residual layers of the branch on it (reference: demucs/demucs.py:133-154 applied to rows after the
// permute of demucs/hdemucs.py:145-151, 316-322):
//     dilated conv3 C->C/8  ->  GroupNorm(1) -> GELU  ->  1x1 C/8->2C  ->  GroupNorm(1)  ->  GLU -> LayerScale -> +x
// GroupNorm(1) of a row needs only that row, so one workgroup can do everything: HBM sees the input row
// (plus L2-served tap / residual re-reads) and the output row, instead of the GEMM route's hidden tensor
// (written once, read three times), residual re-read and separate statistics passes.
// A thread owns 4 adjacent time columns: its (C/8 x 4) hidden values stay in registers, the 1x1 is
// evaluated twice (statistics, then apply) instead of storing its 2C-wide result, activations come in as
// aligned float4 global loads, weights are broadcast from LDS (each ds_read_b128 feeds 16 FMAs).  fp32 VALU
// FMAs: the contraction lengths (3C and C/8) are far too short for MFMA tiles to pay (M = 6 or 12).
#include "common.h"
#include "kernels.h"

namespace mi {

constexpr int kRowThreads = 128;   // 2 waves; T/4 <= 128 column quads

// sum of two doubles over the workgroup; result broadcast to all threads
__device__ __forceinline__ void row_block_sum(double &a, double &b, double *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    const int w = threadIdx.x >> 6;
    __syncthreads();                       // previous users of red[] are done
    if ((threadIdx.x & 63) == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    a = red[0] + red[2]; b = red[1] + red[3];
}

__device__ __forceinline__ float4 ld4(const float *p, bool ok) {
    return ok ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// one residual layer on the row: src -> dst (both [C][T] slices with channel stride cs; may alias)
template <int C, int H, int DIL>
__device__ __forceinline__ void dconv_row_layer(const DConvRowLayer &L, const float *src, float *dst, size_t cs, int T, float *wsm,
                                                double *red) {
    constexpr int HA = (H + 3) / 4 * 4;
    float *w0s = wsm;                       // [C][3][HA]
    float *w3s = w0s + C * 3 * HA;          // [2C][HA]
    float *b3s = w3s + 2 * C * HA;          // [2C]
    float *g2ws = b3s + 2 * C, *g2bs = g2ws + 2 * C;   // [2C] each
    float *lss = g2bs + 2 * C;              // [C]
    float *smalls = lss + C;                // b0, g1w, g1b: [HA] each
    const int tid = threadIdx.x, nq = T >> 2;
    const bool on = tid < nq;
    const int t0 = (on ? tid : 0) * 4;
    __syncthreads();                        // previous layer is done with the LDS weights and with dst
    for (int i = tid; i < C * 3 * HA; i += kRowThreads) w0s[i] = L.w0[i];
    for (int i = tid; i < 2 * C * HA; i += kRowThreads) w3s[i] = L.w3[i];
    for (int i = tid; i < 2 * C; i += kRowThreads) { b3s[i] = L.b3[i]; g2ws[i] = L.g2w[i]; g2bs[i] = L.g2b[i]; }
    for (int i = tid; i < C; i += kRowThreads) lss[i] = L.ls[i];
    if (tid < HA) { smalls[tid] = L.b0[tid]; smalls[HA + tid] = L.g1w[tid]; smalls[2 * HA + tid] = L.g1b[tid]; }
    __syncthreads();

    // ---- dilated conv3 ------------------------------------------------------------------------------
    float hid[4][HA];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < HA; ++m) hid[j][m] = smalls[m];
    const bool has_l = on && t0 >= 4, has_r = on && t0 + 4 < T;
    // activation taps are prefetched one channel ahead (L2 latency would otherwise sit in front of every
    // channel's FMA block)
    float4 na = ld4(src + t0 - 4, has_l), nb = ld4(src + t0, on), nc = ld4(src + t0 + 4, has_r);
#pragma unroll 2
    for (int c = 0; c < C; ++c) {
        const float4 xa = na, xb = nb, xc = nc;
        if (c + 1 < C) {
            const float *p = src + (c + 1) * cs + t0;
            na = ld4(p - 4, has_l); nb = ld4(p, on); nc = ld4(p + 4, has_r);
        }
        const float v[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
        const float4 *wv = reinterpret_cast<const float4 *>(w0s + c * 3 * HA);
#pragma unroll
        for (int q = 0; q < HA / 4; ++q) {
            const float4 wa = wv[q], wb = wv[HA / 4 + q], wc = wv[2 * (HA / 4) + q];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[4 + j - DIL], x1 = v[4 + j], x2 = v[4 + j + DIL];
                hid[j][4 * q + 0] = fmaf(wc.x, x2, fmaf(wb.x, x1, fmaf(wa.x, x0, hid[j][4 * q + 0])));
                hid[j][4 * q + 1] = fmaf(wc.y, x2, fmaf(wb.y, x1, fmaf(wa.y, x0, hid[j][4 * q + 1])));
                hid[j][4 * q + 2] = fmaf(wc.z, x2, fmaf(wb.z, x1, fmaf(wa.z, x0, hid[j][4 * q + 2])));
                hid[j][4 * q + 3] = fmaf(wc.w, x2, fmaf(wb.w, x1, fmaf(wa.w, x0, hid[j][4 * q + 3])));
            }
        }
    }
    // ---- GroupNorm(1, H) over (H, T) + GELU ----------------------------------------------------------
    double s1 = 0.0, s2 = 0.0;
    if (on) {
        float p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < H; ++m) { p1 += hid[j][m]; p2 += hid[j][m] * hid[j][m]; }
        s1 = p1; s2 = p2;
    }
    row_block_sum(s1, s2, red);
    {
        const double cnt = (double)H * T, mean = s1 / cnt;
        const float mu = (float)mean, rs = 1.0f / sqrtf((float)fmax((s2 - s1 * mean) / cnt, 0.0) + 1e-5f);
#pragma unroll
        for (int m = 0; m < HA; ++m) {
            const float gw = smalls[HA + m], gb = smalls[2 * HA + m];
#pragma unroll
            for (int j = 0; j < 4; ++j) hid[j][m] = m < H ? gelu_exact((hid[j][m] - mu) * rs * gw + gb) : 0.f;
        }
    }
    // ---- 1x1 pass A: statistics of z = W3 g + b3 over (2C, T) -----------------------------------------
    s1 = 0.0; s2 = 0.0;
    {
        float p1 = 0.f, p2 = 0.f;
#pragma unroll 2
        for (int m = 0; m < 2 * C; ++m) {
            const float4 *wv = reinterpret_cast<const float4 *>(w3s + m * HA);
            const float bb = b3s[m];
            float z[4] = {bb, bb, bb, bb};
#pragma unroll
            for (int q = 0; q < HA / 4; ++q) {
                const float4 w4 = wv[q];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    z[j] = fmaf(w4.w, hid[j][4 * q + 3], fmaf(w4.z, hid[j][4 * q + 2], fmaf(w4.y, hid[j][4 * q + 1], fmaf(w4.x, hid[j][4 * q], z[j]))));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { p1 += z[j]; p2 += z[j] * z[j]; }
            if ((m & 7) == 7) { if (on) { s1 += p1; s2 += p2; } p1 = 0.f; p2 = 0.f; }   // fp32 partials of 32 values
        }
    }
    row_block_sum(s1, s2, red);
    const double cnt2 = 2.0 * C * T, mean2 = s1 / cnt2;
    const float mu2 = (float)mean2, rs2 = 1.0f / sqrtf((float)fmax((s2 - s1 * mean2) / cnt2, 0.0) + 1e-5f);
    // ---- 1x1 pass B: GroupNorm + GLU + LayerScale + residual --------------------------------------------
    float4 nres = ld4(src + t0, on);
#pragma unroll 2
    for (int c = 0; c < C; ++c) {
        const float4 xr = nres;
        if (c + 1 < C) nres = ld4(src + (c + 1) * cs + t0, on);
        const float4 *wa = reinterpret_cast<const float4 *>(w3s + c * HA);
        const float4 *wg = reinterpret_cast<const float4 *>(w3s + (c + C) * HA);
        const float ba = b3s[c], bg = b3s[c + C];
        float za[4] = {ba, ba, ba, ba}, zg[4] = {bg, bg, bg, bg};
#pragma unroll
        for (int q = 0; q < HA / 4; ++q) {
            const float4 u = wa[q], w = wg[q];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                za[j] = fmaf(u.w, hid[j][4 * q + 3], fmaf(u.z, hid[j][4 * q + 2], fmaf(u.y, hid[j][4 * q + 1], fmaf(u.x, hid[j][4 * q], za[j]))));
                zg[j] = fmaf(w.w, hid[j][4 * q + 3], fmaf(w.z, hid[j][4 * q + 2], fmaf(w.y, hid[j][4 * q + 1], fmaf(w.x, hid[j][4 * q], zg[j]))));
            }
        }
        const float aw = g2ws[c], ab = g2bs[c], gw = g2ws[c + C], gb = g2bs[c + C], sc = lss[c];
        const float r[4] = {xr.x, xr.y, xr.z, xr.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float va = (za[j] - mu2) * rs2 * aw + ab, vg = (zg[j] - mu2) * rs2 * gw + gb;
            o[j] = r[j] + sc * (va * sigmoid_f(vg));
        }
        if (on) *reinterpret_cast<float4 *>(dst + c * cs + t0) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __threadfence_block();                  // the next layer's taps read other threads' columns of dst
}

template <int C, int H>
__global__ __launch_bounds__(kRowThreads) void dconv_row_kernel(const DConvRowArgs a) {
    constexpr int HA = (H + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float wsm[C * 3 * HA + 2 * C * HA + 7 * C + 3 * HA];
    __shared__ double red[4];
    const int row = blockIdx.x, b = row / a.Fr, fr = row - b * a.Fr;
    const size_t cs = (size_t)a.Fr * a.T;
    const size_t off = ((size_t)b * C * a.Fr + fr) * a.T;
    dconv_row_layer<C, H, 1>(a.l[0], a.x + off, a.y + off, cs, a.T, wsm, red);
    dconv_row_layer<C, H, 2>(a.l[1], a.y + off, a.y + off, cs, a.T, wsm, red);
}

bool dconv_row_supported(int C, int T) { return (C == 48 || C == 96) && T % 4 == 0 && T / 4 <= kRowThreads; }

int launch_dconv_row(const DConvRowArgs &a, int C, int rows, hipStream_t st) {
    MI_REQUIRE(dconv_row_supported(C, a.T), "dconv_row: unsupported C=%d T=%d", C, a.T);
    MI_REQUIRE(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.y & 15) == 0, "dconv_row: tensors must be 16-byte aligned");
    if (C == 48) hipLaunchKernelGGL((dconv_row_kernel<48, 6>), dim3(rows), dim3(kRowThreads), 0, st, a);
    else hipLaunchKernelGGL((dconv_row_kernel<96, 12>), dim3(rows), dim3(kRowThreads), 0, st, a);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
