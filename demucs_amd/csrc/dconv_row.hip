// Fused DConv for the frequency branch: one WAVE owns one (batch, frequency-bin) row [C][T] and runs BOTH
// residual layers of the branch on it (reference: demucs/demucs.py:133-154 applied to rows after the permute of
// demucs/hdemucs.py:145-151, 316-322):
//     dilated conv3 C->C/8  ->  GroupNorm(1) -> GELU  ->  1x1 C/8->2C  ->  GroupNorm(1)  ->  GLU -> LayerScale -> +x
// GroupNorm(1) of a row needs only that row, so HBM sees the input row (plus L2-served tap / residual re-reads)
// and the output row, instead of the GEMM route's hidden tensor (written once, read three times), residual
// re-read and separate statistics passes.
// A lane owns 6 adjacent time columns (T = 336 = 56 lanes x 6): its (C/8 x 6) hidden values stay in registers,
// the row statistics are plain wave reductions (no barrier), the 1x1 is evaluated twice (statistics, then
// apply) instead of storing its 2C-wide result, activations come in as aligned float2 loads, weights are
// broadcast from LDS (one ds_read_b128 feeds 24 FMAs) and shared by the 4 rows of a workgroup.
// fp32 VALU FMAs: the contraction lengths (3C and C/8) are far too short for MFMA tiles to pay (M = 6 or 12).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace mi {

// Persistent workgroups: both layers' weights are staged into LDS ONCE per workgroup, then every wave walks rows on its own
// (row = first + k * stride) -- no barrier after the staging, no re-staging per group of four rows.
constexpr int kNC = 6;             // columns per lane

__device__ __forceinline__ void wave_sum2(double &a, double &b) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
}

__device__ __forceinline__ float2 ld2(const float *p, bool ok) {
    return ok ? *reinterpret_cast<const float2 *>(p) : make_float2(0.f, 0.f);
}
// the value lane - 1 / lane + 1 holds in the same register (0 for lane 0 / lane 63): one v_mov_b32_dpp, no LDS
__device__ __forceinline__ float dpp_from_lower(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_upper(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

constexpr int kGramGroup = 16;      // Gram entries reduced per LDS round

// one residual layer on the row: src -> dst (both [C][T] slices with channel stride cs; may alias)
// LDS image of one layer's weights (floats): w0 [C][3][HA], w3 pre-splatted [C][HA][4], b3 / g2w / g2b [2C] each, ls [C], b0 / g1w / g1b [HA] each
template <int C, int H>
__device__ __forceinline__ constexpr int dconv_row_wsm() { return C * 3 * ((H + 3) / 4 * 4) + 4 * C * ((H + 3) / 4 * 4) + 7 * C + 3 * ((H + 3) / 4 * 4); }

template <int C, int H, int NT>
__device__ __forceinline__ void dconv_row_stage(const DConvRowLayer &L, float *wsm) {
    constexpr int HA = (H + 3) / 4 * 4;
    float *w0s = wsm, *w3s = w0s + C * 3 * HA, *b3s = w3s + 4 * C * HA, *g2ws = b3s + 2 * C, *g2bs = g2ws + 2 * C, *lss = g2bs + 2 * C,
          *smalls = lss + C;
    const int tid = threadIdx.x;
    for (int i = tid; i < C * 3 * HA; i += NT) w0s[i] = L.w0[i];
    for (int i = tid; i < 4 * C * HA; i += NT) {
        const int half = (i >> 1) & 1, k = (i >> 2) % HA, c = (i >> 2) / HA;
        w3s[i] = L.w3[(size_t)(c + half * C) * HA + k];
    }
    for (int i = tid; i < 2 * C; i += NT) { b3s[i] = L.b3[i]; g2ws[i] = L.g2w[i]; g2bs[i] = L.g2b[i]; }
    for (int i = tid; i < C; i += NT) lss[i] = L.ls[i];
    if (tid < HA) { smalls[tid] = L.b0[tid]; smalls[HA + tid] = L.g1w[tid]; smalls[2 * HA + tid] = L.g1b[tid]; }
}

template <int C, int H, int DIL>
__device__ __forceinline__ void dconv_row_layer(const DConvTimeLayer &LT, const float *src, float *dst, size_t cs, int T, float *wsm,
                                                float *gpart, bool row_ok) {
    constexpr int HA = (H + 3) / 4 * 4;
    constexpr int UNR = C >= 96 ? 1 : 2;      // channel-loop unrolling: the wider kernel has no registers to spare
    float *w0s = wsm;                       // [C][3][HA]
    float *w3s = w0s + C * 3 * HA;          // [C][HA][4] = (value w, value w, gate w, gate w): pre-splatted v_pk_fma_f32 operands
    float *b3s = w3s + 4 * C * HA;          // [2C]
    float *g2ws = b3s + 2 * C, *g2bs = g2ws + 2 * C;   // [2C] each
    float *lss = g2bs + 2 * C;              // [C]
    float *smalls = lss + C;                // b0, g1w, g1b: [HA] each
    const int tid = threadIdx.x, lane = tid & 63, nq = T / kNC;
    const bool on = row_ok && lane < nq;
    const int t0 = (lane < nq ? lane : 0) * kNC;

    // ---- dilated conv3: taps cover columns t0-2 .. t0+7, fetched as five float2, one channel ahead; hidden channels
    //      in pairs: one v_pk_fma_f32 (two IEEE fmas) per tap and pair ---------------------------------------------
    v2f hid[kNC][HA / 2];
#pragma unroll
    for (int j = 0; j < kNC; ++j)
#pragma unroll
        for (int m = 0; m < HA / 2; ++m) hid[j][m] = (v2f){smalls[2 * m], smalls[2 * m + 1]};
    // A lane fetches only its OWN six columns of a channel (three aligned float2); the two columns on either side that the taps
    // reach come from the neighbouring lanes' registers by DPP wave shifts (v_mov_b32_dpp wave_shr / wave_shl, zero for the lanes
    // without a neighbour = the conv's zero padding at the row ends; lanes past the row hold zeros).  Round 3 loaded all ten
    // columns per lane: 40 instead of 24 bytes per lane and channel through L1 / L2, two thirds more than the row holds.
    // Channels c + 1 .. c + 3 are in flight under the products of channel c (six registers per channel instead of ten).
    float2 a1 = ld2(src + t0, on), a2 = ld2(src + t0 + 2, on), a3 = ld2(src + t0 + 4, on);
    float2 b1 = ld2(src + cs + t0, on), b2 = ld2(src + cs + t0 + 2, on), b3 = ld2(src + cs + t0 + 4, on);
    float2 e1 = ld2(src + 2 * cs + t0, on && C > 2), e2 = ld2(src + 2 * cs + t0 + 2, on && C > 2), e3 = ld2(src + 2 * cs + t0 + 4, on && C > 2);
#pragma unroll UNR
    for (int c = 0; c < C; ++c) {
        const float l0 = dpp_from_lower(a3.x), l1 = dpp_from_lower(a3.y), r0 = dpp_from_upper(a1.x), r1 = dpp_from_upper(a1.y);
        const v2f v[10] = {splat2(l0), splat2(l1), splat2(a1.x), splat2(a1.y), splat2(a2.x), splat2(a2.y), splat2(a3.x), splat2(a3.y),
                           splat2(r0), splat2(r1)};
        a1 = b1; a2 = b2; a3 = b3; b1 = e1; b2 = e2; b3 = e3;
        if (c + 3 < C) {
            const float *p = src + (c + 3) * cs + t0;
            e1 = ld2(p, on); e2 = ld2(p + 2, on); e3 = ld2(p + 4, on);
        }
        const float4 *wv = reinterpret_cast<const float4 *>(w0s + c * 3 * HA);
#pragma unroll
        for (int q = 0; q < HA / 4; ++q) {
            const float4 wa = wv[q], wb = wv[HA / 4 + q], wc = wv[2 * (HA / 4) + q];
#pragma unroll
            for (int j = 0; j < kNC; ++j) {
                const v2f x0 = v[2 + j - DIL], x1 = v[2 + j], x2 = v[2 + j + DIL];
                hid[j][2 * q] = fma2((v2f){wc.x, wc.y}, x2, fma2((v2f){wb.x, wb.y}, x1, fma2((v2f){wa.x, wa.y}, x0, hid[j][2 * q])));
                hid[j][2 * q + 1] = fma2((v2f){wc.z, wc.w}, x2, fma2((v2f){wb.z, wb.w}, x1, fma2((v2f){wa.z, wa.w}, x0, hid[j][2 * q + 1])));
            }
        }
    }
    // ---- GroupNorm(1, H) over (H, T) + GELU ------------------------------------------------------------------
    double s1 = 0.0, s2 = 0.0;
    if (on) {
        float p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int j = 0; j < kNC; ++j)
#pragma unroll
            for (int m = 0; m < H; ++m) { const float hv = hid[j][m >> 1][m & 1]; p1 += hv; p2 += hv * hv; }
        s1 = p1; s2 = p2;
    }
    wave_sum2(s1, s2);
    float g[kNC][HA];
    {
        const double cnt = (double)H * T, mean = s1 / cnt;
        const float mu = (float)mean, rs = 1.0f / sqrtf((float)fmax((s2 - s1 * mean) / cnt, 0.0) + 1e-5f);
#pragma unroll
        for (int m = 0; m < HA; ++m) {
            const float gw = smalls[HA + m], gb = smalls[2 * HA + m];
#pragma unroll
            for (int j = 0; j < kNC; ++j) g[j][m] = (m < H && on) ? gelu_exact((hid[j][m >> 1][m & 1] - mu) * rs * gw + gb) : 0.f;
        }
    }
    // ---- statistics of z = W3 g + b3 over (2C, T) WITHOUT evaluating z: with s = sum_t g[t] and G = sum_t g[t] g[t]^T,
    //      sum z = colsum(W3) . s + T sum(b3),  sum z^2 = <W3^T W3, G> + 2 (W3^T b3) . s + T |b3|^2   (see dconv_time.hip)
    s1 = (double)T * LT.sum_b3; s2 = (double)T * LT.sum_b3sq;
    {
        constexpr int NG = H * (H + 1) / 2 + H;
        // Every lane holds a float32 partial of each of the NG entries (its 6 columns).  Sixteen entries at a time go through this
        // wave's LDS slab [16][64]: lane (entry e, quarter q) adds 16 of the 64 partials in float64, two shuffles join the quarters,
        // and the entry's constants fold it straight into sum z^2 / sum z -- one LDS round per 16 entries instead of a six-level
        // ds_bpermute butterfly per PAIR of entries (45 butterflies at H = 12: a third of the layer's time).
        float *gp = gpart;
        double cq = 0.0, cl = 0.0;
        auto reduce_group = [&](int grp) {
            const int e = lane >> 2, q = lane & 3, n = kGramGroup * grp + e;
            const float4 *src4 = reinterpret_cast<const float4 *>(gp + e * 64 + 16 * q);
            double sacc = 0.0;
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const float4 f = src4[v4];
                sacc += (double)f.x; sacc += (double)f.y; sacc += (double)f.z; sacc += (double)f.w;
            }
            sacc += __shfl_xor(sacc, 1);
            sacc += __shfl_xor(sacc, 2);
            if (q == 0 && n < NG) { cq += LT.gram_e1[n] * sacc; cl += LT.gram_e2[n] * sacc; }
        };
        int n = 0;                                         // compile-time after unrolling
#pragma unroll
        for (int i = 0; i < H; ++i) {
#pragma unroll
            for (int k = i; k <= H; ++k) {                   // k == H: the plain sum of g_i (the vector s)
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < kNC; ++j) p = k < H ? fmaf(g[j][i], g[j][k < H ? k : 0], p) : p + g[j][i];
                gp[(n & (kGramGroup - 1)) * 64 + lane] = p;
                if ((n & (kGramGroup - 1)) == kGramGroup - 1 || n == NG - 1) reduce_group(n / kGramGroup);
                ++n;
            }
        }
        wave_sum2(cq, cl);
        s2 += cq; s1 += cl;
    }
    const double cnt2 = 2.0 * C * T, mean2 = s1 / cnt2;
    const float mu2 = (float)mean2, rs2 = 1.0f / sqrtf((float)fmax((s2 - s1 * mean2) / cnt2, 0.0) + 1e-5f);
    // ---- 1x1: GroupNorm + GLU + LayerScale + residual; adjacent COLUMNS as packed pairs, weights pre-splatted in LDS ------
    v2f gp[kNC / 2][HA];
#pragma unroll
    for (int jp = 0; jp < kNC / 2; ++jp)
#pragma unroll
        for (int m = 0; m < HA; ++m) gp[jp][m] = (v2f){g[2 * jp][m], g[2 * jp + 1][m]};
    float2 r0 = ld2(src + t0, on), r1 = ld2(src + t0 + 2, on), r2 = ld2(src + t0 + 4, on);
    float2 q0 = ld2(src + cs + t0, on), q1 = ld2(src + cs + t0 + 2, on), q2 = ld2(src + cs + t0 + 4, on);
#pragma unroll UNR
    for (int c = 0; c < C; ++c) {
        const v2f r[kNC / 2] = {(v2f){r0.x, r0.y}, (v2f){r1.x, r1.y}, (v2f){r2.x, r2.y}};
        r0 = q0; r1 = q1; r2 = q2;
        if (c + 2 < C) {
            const float *p = src + (c + 2) * cs + t0;
            q0 = ld2(p, on); q1 = ld2(p + 2, on); q2 = ld2(p + 4, on);
        }
        const float4 *wp = reinterpret_cast<const float4 *>(w3s + (size_t)c * HA * 4);
        v2f zv[kNC / 2], zg[kNC / 2];
#pragma unroll
        for (int jp = 0; jp < kNC / 2; ++jp) { zv[jp] = splat2(b3s[c]); zg[jp] = splat2(b3s[c + C]); }
#pragma unroll
        for (int k = 0; k < HA; ++k) {
            const float4 w = wp[k];                      // (value w, value w, gate w, gate w) of hidden channel k
#pragma unroll
            for (int jp = 0; jp < kNC / 2; ++jp) {
                zv[jp] = fma2((v2f){w.x, w.y}, gp[jp][k], zv[jp]);
                zg[jp] = fma2((v2f){w.z, w.w}, gp[jp][k], zg[jp]);
            }
        }
        // GroupNorm affine folded (v = z A + B), GLU, LayerScale, residual: packed pairs of columns; the sigmoid's exponent takes
        // the gate's affine with -log2(e) folded in
        const float aA = rs2 * g2ws[c], aB = g2bs[c] - mu2 * aA, gA = rs2 * g2ws[c + C], gB = g2bs[c + C] - mu2 * gA, sc = lss[c];
        const v2f A2 = splat2(aA), B2 = splat2(aB), G2 = splat2(-1.44269504088896341f * gA), H2 = splat2(-1.44269504088896341f * gB),
                  S2 = splat2(sc), one2 = splat2(1.0f);
        v2f o[kNC / 2];
#pragma unroll
        for (int jp = 0; jp < kNC / 2; ++jp) {
            const v2f val = fma2(zv[jp], A2, B2), ex = fma2(zg[jp], G2, H2);
            const v2f den = (v2f){__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)} + one2;
            const v2f sg = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            o[jp] = fma2(val * sg, S2, r[jp]);
        }
        if (on) {
            float *q = dst + c * cs + t0;
            *reinterpret_cast<float2 *>(q) = make_float2(o[0].x, o[0].y);
            *reinterpret_cast<float2 *>(q + 2) = make_float2(o[1].x, o[1].y);
            *reinterpret_cast<float2 *>(q + 4) = make_float2(o[2].x, o[2].y);
        }
    }
    __threadfence_block();                  // the next layer's taps read other lanes' columns of dst
}

template <int C, int H, int NW>
__global__ __launch_bounds__(64 * NW) void dconv_row_kernel(const DConvRowArgs a, int rows) {
    constexpr int WS = (dconv_row_wsm<C, H>() + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float wsm[2 * WS];
    __shared__ __attribute__((aligned(16))) float gpart[NW * kGramGroup * 64];
    dconv_row_stage<C, H, 64 * NW>(a.l[0].w, wsm);
    dconv_row_stage<C, H, 64 * NW>(a.l[1].w, wsm + WS);
    __syncthreads();
    const size_t cs = (size_t)a.Fr * a.T;
    float *gp = gpart + (threadIdx.x >> 6) * (kGramGroup * 64);
    for (int row = blockIdx.x * NW + (threadIdx.x >> 6); row < rows; row += gridDim.x * NW) {
        const int b = row / a.Fr, fr = row - b * a.Fr;
        const size_t off = ((size_t)b * C * a.Fr + fr) * a.T;
        dconv_row_layer<C, H, 1>(a.l[0], a.x + off, a.y + off, cs, a.T, wsm, gp, true);
        dconv_row_layer<C, H, 2>(a.l[1], a.y + off, a.y + off, cs, a.T, wsm + WS, gp, true);
    }
}

// =====================================================================================================================
// C = 48: the row LIVES IN LDS across both residual layers (read once from HBM, written once).  One 4-wave workgroup per CU and
// row: [C][T] float32 = 64.5 KB next to both layers' weights.  The waves split the CHANNELS, not the columns (a lane keeps its
// six columns, so one broadcast ds_read_b128 of weights still feeds 24 multiply-adds):
//   conv3:   wave w contracts input channels [12 w, 12 w + 12) -> partial hidden values -> LDS -> every wave adds the four
//            partials in a fixed order (identical hidden values, statistics and GELU in all waves: no second exchange);
//   Gram:    the 27 entries that give the second GroupNorm's statistics are dealt out to the waves (entry n -> wave n & 3);
//   1x1:     wave w produces output channels [12 w, 12 w + 12) and adds the residual from the LDS row, in place (layer 1) or
//            straight to HBM (layer 2).
// The next row is fetched into registers (72 per lane; one wave per SIMD has 512) under the arithmetic of the current one.
// The persistent per-wave kernel above re-reads x and the intermediate row through the fabric: 3.0x the algorithmic bytes -- and is
// still the faster one (launch_dconv_row); this kernel is kept selectable (MI_DCONV_ROW=lds) with its parity test.
template <int C, int H, int DIL, bool LAST>
__device__ __forceinline__ void dconv_rowlds_layer(const DConvTimeLayer &LT, float *row, size_t rs /* channel stride of row */, int T, const float *wsm,
                                                   float *hpart, float *gslab, double *gsum, float *out, size_t cs, int wave, int lane) {
    constexpr int HA = (H + 3) / 4 * 4, CW = C / 4;
    const float *w0s = wsm, *w3s = w0s + C * 3 * HA, *b3s = w3s + 4 * C * HA, *g2ws = b3s + 2 * C, *g2bs = g2ws + 2 * C, *lss = g2bs + 2 * C,
                *smalls = lss + C;
    const int nq = T / kNC;
    const bool on = lane < nq;
    const int t0 = (on ? lane : 0) * kNC;
    // ---- dilated conv3 over this wave's input channels (taps from the LDS row) -------------------------------------------------
    v2f hid[kNC][HA / 2];
#pragma unroll
    for (int j = 0; j < kNC; ++j)
#pragma unroll
        for (int m = 0; m < HA / 2; ++m) hid[j][m] = (v2f){0.f, 0.f};
    const bool has_l = t0 >= 2, has_r = t0 + kNC + 2 <= T;
#pragma unroll 2
    for (int cc = 0; cc < CW; ++cc) {
        const int c = wave * CW + cc;
        const float *r = row + c * rs + t0;
        const float2 n0 = ld2(r - 2, has_l), n1 = ld2(r, true), n2 = ld2(r + 2, true), n3 = ld2(r + 4, true), n4 = ld2(r + 6, has_r);
        const v2f v[10] = {splat2(n0.x), splat2(n0.y), splat2(n1.x), splat2(n1.y), splat2(n2.x), splat2(n2.y), splat2(n3.x), splat2(n3.y),
                           splat2(n4.x), splat2(n4.y)};
        const float4 *wv = reinterpret_cast<const float4 *>(w0s + c * 3 * HA);
#pragma unroll
        for (int q = 0; q < HA / 4; ++q) {
            const float4 wa = wv[q], wb = wv[HA / 4 + q], wc = wv[2 * (HA / 4) + q];
#pragma unroll
            for (int j = 0; j < kNC; ++j) {
                const v2f x0 = v[2 + j - DIL], x1 = v[2 + j], x2 = v[2 + j + DIL];
                hid[j][2 * q] = fma2((v2f){wc.x, wc.y}, x2, fma2((v2f){wb.x, wb.y}, x1, fma2((v2f){wa.x, wa.y}, x0, hid[j][2 * q])));
                hid[j][2 * q + 1] = fma2((v2f){wc.z, wc.w}, x2, fma2((v2f){wb.z, wb.w}, x1, fma2((v2f){wa.z, wa.w}, x0, hid[j][2 * q + 1])));
            }
        }
    }
    {
        float *hp = hpart + wave * (kNC * H) * 64 + lane;
#pragma unroll
        for (int j = 0; j < kNC; ++j)
#pragma unroll
            for (int m = 0; m < H; ++m) hp[(j * H + m) * 64] = hid[j][m >> 1][m & 1];
    }
    __syncthreads();
    float hv[kNC][H];
#pragma unroll
    for (int j = 0; j < kNC; ++j)
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const float *hp = hpart + (j * H + m) * 64 + lane;
            hv[j][m] = (((smalls[m] + hp[0]) + hp[(kNC * H) * 64]) + hp[2 * (kNC * H) * 64]) + hp[3 * (kNC * H) * 64];
        }
    // ---- GroupNorm(1, H) over (H, T) + GELU: every wave holds the whole hidden row ------------------------------------------------
    double s1 = 0.0, s2 = 0.0;
    if (on) {
        float p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int j = 0; j < kNC; ++j)
#pragma unroll
            for (int m = 0; m < H; ++m) { p1 += hv[j][m]; p2 += hv[j][m] * hv[j][m]; }
        s1 = p1; s2 = p2;
    }
    wave_sum2(s1, s2);
    float g[kNC][HA];
    {
        const double cnt = (double)H * T, mean = s1 / cnt;
        const float mu = (float)mean, rs = 1.0f / sqrtf((float)fmax((s2 - s1 * mean) / cnt, 0.0) + 1e-5f);
#pragma unroll
        for (int m = 0; m < HA; ++m) {
            const float gw = smalls[HA + (m < H ? m : 0)], gb = smalls[2 * HA + (m < H ? m : 0)];
#pragma unroll
            for (int j = 0; j < kNC; ++j) g[j][m] = (m < H && on) ? gelu_exact((hv[j][m < H ? m : 0] - mu) * rs * gw + gb) : 0.f;
        }
    }
    // ---- second GroupNorm's statistics from the Gram matrix (see dconv_row_layer); entry n belongs to wave n & 3 -----------------
    {
        constexpr int NG = H * (H + 1) / 2 + H;
        static_assert((NG + 3) / 4 <= 8, "gram slab holds eight entries per wave");
        float *gp = gslab + wave * (8 * 64);
        int n = 0;                                         // compile-time after unrolling
#pragma unroll
        for (int i = 0; i < H; ++i) {
#pragma unroll
            for (int k = i; k <= H; ++k) {                   // k == H: the plain sum of g_i
                if ((n & 3) == wave) {
                    float p = 0.f;
#pragma unroll
                    for (int j = 0; j < kNC; ++j) p = k < H ? fmaf(g[j][i], g[j][k < H ? k : 0], p) : p + g[j][i];
                    gp[(n >> 2) * 64 + lane] = p;
                }
                ++n;
            }
        }
        // lane (entry slot e, quarter q) adds 16 of the 64 partials in float64; two shuffles join the quarters
        const int e = lane >> 2, q = lane & 3, ne = 4 * e + wave;
        double cq = 0.0, cl = 0.0;
        if (e < 8) {
            const float4 *src4 = reinterpret_cast<const float4 *>(gp + e * 64 + 16 * q);
            double sacc = 0.0;
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const float4 f = src4[v4];
                sacc += (double)f.x; sacc += (double)f.y; sacc += (double)f.z; sacc += (double)f.w;
            }
            sacc += __shfl_xor(sacc, 1);
            sacc += __shfl_xor(sacc, 2);
            if (q == 0 && ne < NG) { cq = LT.gram_e1[ne] * sacc; cl = LT.gram_e2[ne] * sacc; }
        }
        wave_sum2(cq, cl);
        if (lane == 0) { gsum[2 * wave] = cq; gsum[2 * wave + 1] = cl; }
        __syncthreads();
        s1 = (double)T * LT.sum_b3; s2 = (double)T * LT.sum_b3sq;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s2 += gsum[2 * w]; s1 += gsum[2 * w + 1]; }
    }
    const double cnt2 = 2.0 * C * T, mean2 = s1 / cnt2;
    const float mu2 = (float)mean2, rs2 = 1.0f / sqrtf((float)fmax((s2 - s1 * mean2) / cnt2, 0.0) + 1e-5f);
    // ---- 1x1 + GroupNorm + GLU + LayerScale + residual for this wave's output channels ------------------------------------------
    v2f gp2[kNC / 2][HA];
#pragma unroll
    for (int jp = 0; jp < kNC / 2; ++jp)
#pragma unroll
        for (int m = 0; m < HA; ++m) gp2[jp][m] = (v2f){g[2 * jp][m], g[2 * jp + 1][m]};
#pragma unroll 2
    for (int cc = 0; cc < CW; ++cc) {
        const int c = wave * CW + cc;
        float *rp = row + c * rs + t0;
        const float2 r0 = ld2(rp, true), r1 = ld2(rp + 2, true), r2 = ld2(rp + 4, true);
        const v2f r[kNC / 2] = {(v2f){r0.x, r0.y}, (v2f){r1.x, r1.y}, (v2f){r2.x, r2.y}};
        const float4 *wp = reinterpret_cast<const float4 *>(w3s + (size_t)c * HA * 4);
        v2f zv[kNC / 2], zg[kNC / 2];
#pragma unroll
        for (int jp = 0; jp < kNC / 2; ++jp) { zv[jp] = splat2(b3s[c]); zg[jp] = splat2(b3s[c + C]); }
#pragma unroll
        for (int k = 0; k < HA; ++k) {
            const float4 w = wp[k];                      // (value w, value w, gate w, gate w) of hidden channel k
#pragma unroll
            for (int jp = 0; jp < kNC / 2; ++jp) {
                zv[jp] = fma2((v2f){w.x, w.y}, gp2[jp][k], zv[jp]);
                zg[jp] = fma2((v2f){w.z, w.w}, gp2[jp][k], zg[jp]);
            }
        }
        const float aA = rs2 * g2ws[c], aB = g2bs[c] - mu2 * aA, gA = rs2 * g2ws[c + C], gB = g2bs[c + C] - mu2 * gA, sc = lss[c];
        const v2f A2 = splat2(aA), B2 = splat2(aB), G2 = splat2(-1.44269504088896341f * gA), H2 = splat2(-1.44269504088896341f * gB),
                  S2 = splat2(sc), one2 = splat2(1.0f);
        v2f o[kNC / 2];
#pragma unroll
        for (int jp = 0; jp < kNC / 2; ++jp) {
            const v2f val = fma2(zv[jp], A2, B2), ex = fma2(zg[jp], G2, H2);
            const v2f den = (v2f){__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)} + one2;
            const v2f sg = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            o[jp] = fma2(val * sg, S2, r[jp]);
        }
        if (on) {
            float *q = LAST ? out + c * cs + t0 : rp;
            *reinterpret_cast<float2 *>(q) = make_float2(o[0].x, o[0].y);
            *reinterpret_cast<float2 *>(q + 2) = make_float2(o[1].x, o[1].y);
            *reinterpret_cast<float2 *>(q + 4) = make_float2(o[2].x, o[2].y);
        }
    }
    if (!LAST) __syncthreads();             // the next layer's taps read every channel of the row
}

constexpr int kRowLdsT = 64 * kNC;          // longest row (columns) the LDS-resident kernel takes

template <int C, int H>
__global__ __launch_bounds__(256) void dconv_rowlds_kernel(const DConvRowArgs a, int rows) {
    constexpr int WS = (dconv_row_wsm<C, H>() + 3) / 4 * 4, CW = C / 4;
    __shared__ __attribute__((aligned(16))) float wsm[2 * WS];
    __shared__ __attribute__((aligned(16))) float rowl[8 + C * kRowLdsT];   // 8 floats in front: the left taps of channel 0 stay inside
    __shared__ __attribute__((aligned(16))) float hpart[4 * kNC * H * 64];
    __shared__ __attribute__((aligned(16))) float gslab[4 * 8 * 64];
    __shared__ double gsum[8];
    dconv_row_stage<C, H, 256>(a.l[0].w, wsm);
    dconv_row_stage<C, H, 256>(a.l[1].w, wsm + WS);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = a.T, nq = T / kNC;
    const bool on = lane < nq;
    const int t0 = (on ? lane : 0) * kNC;
    const size_t cs = (size_t)a.Fr * T;
    float *row = rowl + 8;
    float2 pf[CW][3];
    auto fetch = [&](int r) {
        const int b = r / a.Fr, fr = r - b * a.Fr;
        const float *p = a.x + ((size_t)b * C * a.Fr + fr) * T + (size_t)(wave * CW) * cs + t0;
#pragma unroll
        for (int cc = 0; cc < CW; ++cc) {
            pf[cc][0] = ld2(p + cc * cs, on); pf[cc][1] = ld2(p + cc * cs + 2, on); pf[cc][2] = ld2(p + cc * cs + 4, on);
        }
    };
    int r = blockIdx.x;
    if (r < rows) fetch(r);
    for (; r < rows; r += gridDim.x) {
        if (on) {
#pragma unroll
            for (int cc = 0; cc < CW; ++cc) {
                float *q = row + (wave * CW + cc) * T + t0;
                *reinterpret_cast<float2 *>(q) = pf[cc][0];
                *reinterpret_cast<float2 *>(q + 2) = pf[cc][1];
                *reinterpret_cast<float2 *>(q + 4) = pf[cc][2];
            }
        }
        __syncthreads();                    // the row (and, the first time, the weights) are in LDS
        if (r + (int)gridDim.x < rows) fetch(r + gridDim.x);
        const int b = r / a.Fr, fr = r - b * a.Fr;
        float *out = a.y + ((size_t)b * C * a.Fr + fr) * T;
        dconv_rowlds_layer<C, H, 1, false>(a.l[0], row, (size_t)T, T, wsm, hpart, gslab, gsum, nullptr, cs, wave, lane);
        dconv_rowlds_layer<C, H, 2, true>(a.l[1], row, (size_t)T, T, wsm + WS, hpart, gslab, gsum, out, cs, wave, lane);
    }
}

bool dconv_row_supported(int C, int T) { return (C == 48 || C == 96) && T % kNC == 0 && T % 2 == 0 && T / kNC <= 64; }

int launch_dconv_row(const DConvRowArgs &a, int C, int rows, hipStream_t st) {
    MI_REQUIRE(dconv_row_supported(C, a.T), "dconv_row: unsupported C=%d T=%d", C, a.T);
    MI_REQUIRE(((uintptr_t)a.x & 7) == 0 && ((uintptr_t)a.y & 7) == 0, "dconv_row: tensors must be 8-byte aligned");
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    // C = 48: 4-wave workgroups (40 KiB of LDS: three per CU by LDS, registers allow three waves per SIMD);
    // C = 96: 8-wave workgroups (102 KiB: one per CU, two waves per SIMD as the 200 registers allow)
    // MI_DCONV_ROW=lds selects the LDS-resident kernel (1.0x the algorithmic HBM bytes, but 1.87 ms per level-0 launch against
    // 1.19 ms: one wave per SIMD exposes every LDS / barrier latency, and statistics + GELU are repeated in all four waves)
    static const bool lds_row = [] { const char *e = getenv("MI_DCONV_ROW"); return e && e[0] == 'l'; }();
    if (C == 48 && lds_row && a.T <= kRowLdsT) {
        hipLaunchKernelGGL((dconv_rowlds_kernel<48, 6>), dim3(std::min(rows, cus)), dim3(256), 0, st, a, rows);
    } else if (C == 48) {
        const int nblk = std::min(ceil_div(rows, 4), 3 * cus);
        hipLaunchKernelGGL((dconv_row_kernel<48, 6, 4>), dim3(nblk), dim3(256), 0, st, a, rows);
    } else {
        const int nblk = std::min(ceil_div(rows, 8), cus);
        hipLaunchKernelGGL((dconv_row_kernel<96, 12, 8>), dim3(nblk), dim3(512), 0, st, a, rows);
    }
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
