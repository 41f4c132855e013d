// Kernels of the Hybrid Demucs v3 (`hdemucs_mmi`) path that the htdemucs engine has no counterpart for
// (reference: demucs/hdemucs.py:92-157,304-335 GroupNorm(4) inside the layers; demucs/demucs.py:20-67 BLSTM with
// overlapping 200-step chunks; demucs/demucs.py:182-216 LocalState attention).  All float32 (matrix-pipe products are the exact
// float32 MFMA forms).
#include "common.h"
#include "kernels.h"

namespace mi {

// ---- y[b][c][0:L] (row pitch out_pitch) = (x[b][c][0:L] - mean[b]) * inv[b]; padding columns are zeroed --------------
__global__ __launch_bounds__(256) void row_affine_pitch_kernel(const float *__restrict__ x, int L, int out_pitch, int C,
                                                               const float2 *__restrict__ norm, float *__restrict__ y) {
    const int bc = blockIdx.y, b = bc / C;
    const float2 nm = norm[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < out_pitch; i += gridDim.x * 256)
        y[(size_t)bc * out_pitch + i] = i < L ? (x[(size_t)bc * L + i] - nm.x) * nm.y : 0.f;
}

int launch_row_affine_pitch(const float *x, int B, int C, int L, int out_pitch, const float2 *norm, float *y, hipStream_t st) {
    hipLaunchKernelGGL(row_affine_pitch_kernel, dim3(std::min(1024, ceil_div(out_pitch, 256)), B * C), dim3(256), 0, st, x, L, out_pitch, C, norm, y);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ---- GroupNorm apply with the activations / gating / residual forms the layers use ---------------------------------
//   x (B, Cin, in_len) with row pitch in_pitch, statistics per (b, group) with Cin / G channels per group;
//   y[b][co][p] = res? + scale? * act( n(co, p + off) [* sigmoid(n(co + Cout, p + off)) if glu] ),  p in [0, out_len)
//   n(c, q) = (x[b][c][q] - mean) * rstd * w[c] + bias[c]
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, int Cin, int G, int in_pitch, int off,
                                                       const float2 *__restrict__ stats, const float *__restrict__ w,
                                                       const float *__restrict__ bias, int glu, int gelu,
                                                       const float *__restrict__ scale, const float *__restrict__ res, int res_pitch,
                                                       float *__restrict__ y, int Cout, int out_len, int out_pitch, int chan_div) {
    // chan_div > 1: the "channels" here are (channel, row) pairs of a (C, rows, len) tensor; the affine is per real channel
    const int co = blockIdx.y, b = blockIdx.z;
    const int cg = Cin / G;
    const float2 sa = stats[b * G + co / cg];
    const float aA = sa.y * w[co / chan_div], aB = bias[co / chan_div] - sa.x * aA;
    float gA = 0.f, gB = 0.f;
    if (glu) {
        const float2 sg = stats[b * G + (co + Cout) / cg];
        gA = sg.y * w[(co + Cout) / chan_div]; gB = bias[(co + Cout) / chan_div] - sg.x * gA;
    }
    const float *xa = x + ((size_t)b * Cin + co) * in_pitch + off;
    const float *xg = x + ((size_t)b * Cin + co + Cout) * in_pitch + off;
    const float sc = scale ? scale[co / chan_div] : 1.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < out_len; p += gridDim.x * 256) {
        float v = fmaf(xa[p], aA, aB);
        if (glu) v *= sigmoid_f(fmaf(xg[p], gA, gB));
        if (gelu) v = gelu_exact(v);
        v *= sc;
        if (res) v += res[((size_t)b * Cout + co) * res_pitch + p];
        y[((size_t)b * Cout + co) * out_pitch + p] = v;
    }
}

int launch_gn_apply(const float *x, int B, int Cin, int G, int in_pitch, int off, const float2 *stats, const float *w, const float *bias,
                    int glu, int gelu, const float *scale, const float *res, int res_pitch, float *y, int Cout, int out_len,
                    int out_pitch, hipStream_t st, int chan_div) {
    MI_REQUIRE(Cin % G == 0 && (!glu || Cin == 2 * Cout) && (glu || Cin == Cout), "gn_apply: bad channel counts %d -> %d", Cin, Cout);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(std::max(1, std::min(64, ceil_div(out_len, 256))), Cout, B), dim3(256), 0, st, x, Cin, G, in_pitch, off,
                       stats, w, bias, glu, gelu, scale, res, res_pitch, y, Cout, out_len, out_pitch, chan_div);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ---- BLSTM framing (demucs/utils.py:20-35, demucs/demucs.py:38-44,51-64) ---------------------------------------------
// x (B, C, T) -> frames (B * F, C, W): frame f = columns [f * S, f * S + W), zero past T
__global__ __launch_bounds__(256) void unfold_frames_kernel(const float *__restrict__ x, int C, int T, int F, int W, int S,
                                                            float *__restrict__ fr) {
    const int n = blockIdx.z, c = blockIdx.y, b = n / F, f = n % F;
    for (int w = blockIdx.x * 256 + threadIdx.x; w < W; w += gridDim.x * 256) {
        const int t = f * S + w;
        fr[((size_t)n * C + c) * W + w] = t < T ? x[((size_t)b * C + c) * T + t] : 0.f;
    }
}
// frames (B * F, C, W) -> y (B, C, T) = skip + the kept part of each frame: frame 0 keeps [0, W - S/2), the last one
// [S/2, W), the others [S/2, W - S/2); concatenated and cut to T
__global__ __launch_bounds__(256) void restitch_frames_kernel(const float *__restrict__ fr, int C, int T, int F, int W, int S,
                                                              const float *__restrict__ skip, float *__restrict__ y) {
    const int b = blockIdx.z, c = blockIdx.y, lim = S / 2;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
        // output position t lies in frame f at column w: frame 0 covers t in [0, W - lim); frame f >= 1 starts at
        // (W - lim) + (f - 1) * (W - 2 lim) and maps to w = lim + (t - start)
        int f, w;
        if (t < W - lim || F == 1) { f = 0; w = t; }
        else {
            const int u = t - (W - lim), per = W - 2 * lim;
            f = 1 + u / per; w = lim + u % per;
            if (f > F - 1) { w += (f - (F - 1)) * per; f = F - 1; }      // the last frame keeps everything to its end
        }
        const size_t o = ((size_t)b * C + c) * T + t;
        y[o] = fr[(((size_t)b * F + f) * C + c) * W + w] + (skip ? skip[o] : 0.f);
    }
}

int launch_unfold_frames(const float *x, int B, int C, int T, int F, int W, int S, float *fr, hipStream_t st) {
    hipLaunchKernelGGL(unfold_frames_kernel, dim3(1, C, B * F), dim3(256), 0, st, x, C, T, F, W, S, fr);
    MI_CHECK_LAUNCH();
    return MI_OK;
}
int launch_restitch_frames(const float *fr, int B, int C, int T, int F, int W, int S, const float *skip, float *y, hipStream_t st) {
    hipLaunchKernelGGL(restitch_frames_kernel, dim3(std::max(1, std::min(16, ceil_div(T, 256))), C, B), dim3(256), 0, st, fr, C, T, F, W, S, skip, y);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ---- LSTM recurrence (nn.LSTM semantics, zero initial state, gate order i, f, g, o) ------------------------------------
// gx (N, 2 dirs, 4H, W): W_ih x_t + b_ih + b_hh for every step (a GEMM done before); out (N, 2H, W): forward hidden states
// in channels [0, H), backward ones in [H, 2H).
// ONE LAUNCH PER TIME STEP: the recurrent matrix (2.4 MB per direction at H = 384) cannot stream through one CU per step
// (a persistent workgroup per sequence ran at 36 us per step), so every step spreads it over (H / 4) x 2 x ceil(N / 32)
// workgroups: a workgroup owns 4 hidden units (their 16 gate rows) of one direction for a tile of 32 sequences and
// computes the 16 x 32 block  W_hh[rows, :] . h_prev[:, tile]  on the matrix pipe -- v_mfma_f32_16x16x4_f32, exact
// float32 products; each of the 4 waves takes a quarter of the k range (its weights stay in 12 / 24 registers, the
// previous hidden states come from an LDS image of the tile) and the four partial blocks are summed by the gate threads.
// whh is packed for that (pack_lstm_whh below): [dir][H / 4 unit blocks][wave][k step / 4][lane][4], the element of lane l
// at k step i being W_hh[gate (l % 16) / 4, unit (l % 16) % 4][k = wave * H / 4 + 4 i + l / 16].
// The hidden state ping-pongs between two global buffers; the launch boundary is the step barrier.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int H>
__global__ __launch_bounds__(256) void lstm_step_kernel(const float *__restrict__ gx, const float *__restrict__ whh,
                                                        const float *__restrict__ hprev, float *__restrict__ hnext,
                                                        float *__restrict__ cst, float *__restrict__ out, int N, int W, int step) {
    constexpr int RB = 4, NT = 32, KW = H / 4, NK = KW / 4;
    __shared__ float hs[NT][H + 4];
    __shared__ float part[4][4 * RB][NT + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, dir = blockIdx.y, jb = blockIdx.x, j0 = jb * RB;
    const int t = dir ? W - 1 - step : step;
    const int n0 = blockIdx.z * NT, nn = min(NT, N - n0);
    // the gate threads fetch their pre-activations and cell state first: the strided reads land while the products run
    const bool gate_thread = tid < RB * nn;
    const int gu = tid % RB, gnl = tid / RB, gn = n0 + gnl, gj = j0 + gu;
    const size_t si = ((size_t)dir * N + gn) * H + gj;
    float pi = 0.f, pf = 0.f, pg = 0.f, po = 0.f, pc = 0.f;
    if (gate_thread) {
        const float *g = gx + (((size_t)gn * 2 + dir) * 4 * H) * W + t;
        pi = g[(size_t)(0 * H + gj) * W]; pf = g[(size_t)(1 * H + gj) * W];
        pg = g[(size_t)(2 * H + gj) * W]; po = g[(size_t)(3 * H + gj) * W];
        pc = cst[si];
    }
    float a[NK];
    const float4 *wp = (const float4 *)(whh + (((size_t)dir * (H / RB) + jb) * 4 + wave) * NK * 64) + lane;
#pragma unroll
    for (int q = 0; q < NK / 4; ++q) {
        const float4 v = wp[q * 64];
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
    const float4 *hp4 = (const float4 *)(hprev + ((size_t)dir * N + n0) * H);
    for (int i = tid; i < NT * (H / 4); i += 256) {          // rows beyond the tile's sequences read as zero
        const int n = i / (H / 4);
        *(float4 *)&hs[n][(i % (H / 4)) * 4] = n < nn ? hp4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    v4f c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    const float *b0 = &hs[lane & 15][wave * KW + (lane >> 4)], *b1 = &hs[16 + (lane & 15)][wave * KW + (lane >> 4)];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b0[4 * i], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b1[4 * i], c1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                             // D[row 4 (lane / 16) + r][col lane % 16]
        part[wave][4 * (lane >> 4) + r][lane & 15] = c0[r];
        part[wave][4 * (lane >> 4) + r][16 + (lane & 15)] = c1[r];
    }
    __syncthreads();
    if (gate_thread) {
        float s4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            s4[g] = (part[0][g * RB + gu][gnl] + part[1][g * RB + gu][gnl]) + (part[2][g * RB + gu][gnl] + part[3][g * RB + gu][gnl]);
        float c = pc;
        const float h = lstm_cell(pi + s4[0], pf + s4[1], pg + s4[2], po + s4[3], c, lstm_tag(step));
        cst[si] = c;
        hnext[si] = h;
        out[((size_t)gn * 2 * H + dir * H + gj) * W + t] = h;
    }
}

// W_hh of both directions (2, 4H, H) -> the step kernel's operand order; `packed` holds 2 * 4H * H floats
void pack_lstm_whh(const float *whh, int H, float *packed) {
    const int KW = H / 4, NK = KW / 4;
    for (int dir = 0; dir < 2; ++dir)
        for (int jb = 0; jb < H / 4; ++jb)
            for (int wave = 0; wave < 4; ++wave)
                for (int i = 0; i < NK; ++i)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int r = lane & 15, row = (r / 4) * H + jb * 4 + r % 4, k = wave * KW + 4 * i + (lane >> 4);
                        packed[((((size_t)dir * (H / 4) + jb) * 4 + wave) * NK + (i / 4) * 4) * 64 + lane * 4 + i % 4] =
                            whh[((size_t)dir * 4 * H + row) * H + k];
                    }
}

// ---- any other (small) hidden size: the reference's own `demucs_unittest` model is HDemucs(channels=4) (pretrained.py:27-29), whose
// layers 4 / 5 have H = 16 / 32.  One workgroup per (sequence, direction) walks the W steps with W_hh (natural (2, 4H, H) order) and
// h in LDS; thread j < 4H owns gate row j.  Plain fmaf chains in k order: not a performance path.
__global__ __launch_bounds__(256) void lstm_small_kernel(const float *__restrict__ gx, const float *__restrict__ whh, float *__restrict__ out, int N, int H,
                                                         int W) {
    extern __shared__ float lsm[];
    float *wl = lsm, *hs = wl + 4 * H * H, *gs = hs + H;
    const int n = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    for (int i = j; i < 4 * H * H; i += 256) wl[i] = whh[(size_t)dir * 4 * H * H + i];
    if (j < H) hs[j] = 0.f;
    float c = 0.f;
    __syncthreads();
    for (int s = 0; s < W; ++s) {
        const int t = dir ? W - 1 - s : s;
        if (j < 4 * H) {
            float a = 0.f;
            for (int k = 0; k < H; ++k) a = fmaf(wl[j * H + k], hs[k], a);
            gs[j] = gx[(((size_t)n * 2 + dir) * 4 * H + j) * W + t] + a;
        }
        __syncthreads();
        if (j < H) {
            const float h = lstm_cell(gs[j], gs[H + j], gs[2 * H + j], gs[3 * H + j], c, lstm_tag(s));
            hs[j] = h;
            out[((size_t)n * 2 * H + dir * H + j) * W + t] = h;
        }
        __syncthreads();
    }
}

int launch_lstm_small(const float *gx, const float *whh_natural, int N, int H, int W, float *out, hipStream_t st) {
    MI_REQUIRE(H >= 1 && H <= 64, "lstm: hidden size %d has no kernel (192 / 384: matrix-pipe kernels; <= 64: the generic one)", H);
    hipLaunchKernelGGL(lstm_small_kernel, dim3(N, 2), dim3(256), (size_t)(4 * H * H + 5 * H) * sizeof(float), st, gx, whh_natural, out, N, H, W);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// state: 3 buffers of 2 x N x H floats (h ping, h pong, c), zeroed here
int launch_lstm_seq(const float *gx, const float *whh, int N, int H, int W, float *out, float *state, hipStream_t st) {
    MI_REQUIRE(H == 192 || H == 384, "lstm: hidden size %d not instantiated", H);
    const size_t sz = (size_t)2 * N * H;
    MI_HIP(hipMemsetAsync(state, 0, 3 * sz * sizeof(float), st));
    float *h0 = state, *h1 = state + sz, *c = state + 2 * sz;
    const dim3 grid(H / 4, 2, ceil_div(N, 32));
    for (int s = 0; s < W; ++s) {
        if (H == 192) hipLaunchKernelGGL(lstm_step_kernel<192>, grid, dim3(256), 0, st, gx, whh, h0, h1, c, out, N, W, s);
        else hipLaunchKernelGGL(lstm_step_kernel<384>, grid, dim3(256), 0, st, gx, whh, h0, h1, c, out, N, W, s);
        std::swap(h0, h1);
    }
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ---- LocalState attention (demucs/demucs.py:182-216) ------------------------------------------------------------------
// qkc (B, 3C + 16, ld): rows [0, C) queries, [C, 2C) keys, [2C, 3C) content, [3C, 3C + 16) decay logits (heads x 4); the row
// pitch ld is a multiple of 4 (>= T).  For query s of head h:
//     score(t) = k_t . q_s / sqrt(dh) - |t - s| * slope_s,  slope_s = sum_f (f + 1) * (sigmoid(d_f) / 2) / 2,  score(s) = -100,
// softmax over t, out[c][s] = sum_t w_t content[c][t].
// Flash attention on the float32 matrix pipe, the scheme of attention.hip: a wave owns 32 queries and computes the TRANSPOSED
// score tile S^T[key][query] = K^T Q per 32 keys (keys on MFMA rows, queries on lanes), adds the distance penalty and the
// diagonal fill in registers, and after the online softmax its 16 accumulator registers ARE the B operand of
// O^T[d][query] += content[d][key] P^T[key][query].  Workgroup = 2 waves = 64 queries sharing the K / content LDS tiles
// (64 keys); DH = 48 runs its output product on two 32-row tiles with rows 48..63 zero.
template <int DH>
__global__ __launch_bounds__(128) void local_attn_kernel(const float *__restrict__ qkc, int C, int T, int ld, float *__restrict__ out,
                                                         int ld_o) {
    constexpr int KT = 64, VLD = KT + 1, ND = (DH + 31) / 32, HD = ND * 32, NS = DH / 2;
    __shared__ float Ks[DH][KT];
    __shared__ float Vs[HD][VLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 64 + wave * 32;
    const float *base = qkc + (size_t)b * (3 * C + 16) * ld;
    const float *qp = base + (size_t)head * DH * ld, *kp = base + (size_t)(C + head * DH) * ld,
                *vp = base + (size_t)(2 * C + head * DH) * ld, *dp = base + (size_t)(3 * C + head * 4) * ld;
    const int qi = q0 + li;
    const bool qok = qi < T;
    const float isq = 1.0f / sqrtf((float)DH);
    float qreg[NS];                                       // B operand of S^T: lane (query li, half lh) holds Q[d = 2 s + lh][qi] / sqrt(dh)
#pragma unroll
    for (int s = 0; s < NS; ++s) qreg[s] = qok ? qp[(size_t)(2 * s + lh) * ld + qi] * isq : 0.f;
    float slope = 0.f;
    if (qok) {
#pragma unroll
        for (int f = 0; f < 4; ++f) slope += (float)(f + 1) * (sigmoid_f(dp[(size_t)f * ld + qi]) * 0.5f);
    }
    slope *= 0.5f;                                        // / ndecay ** 0.5
    f32x16 oacc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    if (HD > DH) {                                        // zero rows of the padded content tile, written once
        for (int i = tid; i < (HD - DH) * VLD; i += 128) (&Vs[DH][0])[i] = 0.f;
    }
    const int sr = tid >> 4, sc4 = (tid & 15) * 4;        // tile loads: rows sr + 8 it, columns sc4 .. sc4 + 3
    for (int k0 = 0; k0 < T; k0 += KT) {
        __syncthreads();                                  // previous tile fully consumed
#pragma unroll 4
        for (int r = sr; r < DH; r += 8) {
            float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
            if (k0 + sc4 + 3 < T) {
                kk = *reinterpret_cast<const float4 *>(kp + (size_t)r * ld + k0 + sc4);
                vv = *reinterpret_cast<const float4 *>(vp + (size_t)r * ld + k0 + sc4);
            } else {                                      // ragged end: the pitch padding holds no defined values
                float tk[4] = {0.f, 0.f, 0.f, 0.f}, tv[4] = {0.f, 0.f, 0.f, 0.f};
                for (int e = 0; e < 4; ++e)
                    if (k0 + sc4 + e < T) { tk[e] = kp[(size_t)r * ld + k0 + sc4 + e]; tv[e] = vp[(size_t)r * ld + k0 + sc4 + e]; }
                kk = make_float4(tk[0], tk[1], tk[2], tk[3]); vv = make_float4(tv[0], tv[1], tv[2], tv[3]);
            }
            *reinterpret_cast<float4 *>(&Ks[r][sc4]) = kk;
            Vs[r][sc4] = vv.x; Vs[r][sc4 + 1] = vv.y; Vs[r][sc4 + 2] = vv.z; Vs[r][sc4 + 3] = vv.w;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kb = sub * 32;
            if (k0 + kb >= T) break;                      // workgroup-uniform
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            float kf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = Ks[2 * s + lh][kb + li];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float a = kf[s & 3];
                if (s + 4 < NS) kf[s & 3] = Ks[2 * (s + 4) + lh][kb + li];
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[s], sacc, 0, 0, 0);
            }
            float vf[ND][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kc = kb + (s & 3) + 8 * (s >> 2) + 4 * lh;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) vf[dt][s] = Vs[dt * 32 + li][kc];
            }
            // register r of lane (li, lh) is key k0 + kb + (r & 3) + 8 (r >> 2) + 4 lh, query qi
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = k0 + kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float sc = sacc[r] - fabsf((float)(t - qi)) * slope;
                if (t == qi) sc = -100.f;
                if (t >= T) sc = -INFINITY;
                sacc[r] = sc;
                mloc = fmaxf(mloc, sc);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float mnew = fmaxf(mrun, mloc);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(sacc[r] - mnew);
                sacc[r] = p;
                psum += p;
            }
            if (__any(mnew != mrun)) {
                const float alpha = __expf(mrun - mnew);  // exp(-inf) = 0 on the first tile
                lrun *= alpha;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                mrun = mnew;
            }
            lrun += psum;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float a[ND];
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) a[dt] = vf[dt][s & 3];
                if (s + 4 < 16) {
                    const int kc = kb + ((s + 4) & 3) + 8 * ((s + 4) >> 2) + 4 * lh;
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) vf[dt][s & 3] = Vs[dt * 32 + li][kc];
                }
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt], sacc[s], oacc[dt], 0, 0, 0);
            }
        }
    }
    const float ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.0f / ltot;
    if (qok) {
        float *op = out + ((size_t)b * C + head * DH) * ld_o + qi;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dd = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (dd < DH) op[(size_t)dd * ld_o] = oacc[dt][r] * inv;
            }
    }
}

// LocalState for any other width (head dimension C / 4 <= 16: demucs_unittest's 4 / 8): one thread per (item, head, query) with an
// online softmax over all keys, same definition as above
__global__ __launch_bounds__(256) void local_attn_small_kernel(const float *__restrict__ qkc, int C, int T, int ld, float *__restrict__ out, int ld_o) {
    const int s = blockIdx.x * 256 + threadIdx.x, head = blockIdx.y, b = blockIdx.z, dh = C / 4;
    if (s >= T) return;
    const float *base = qkc + (size_t)b * (3 * C + 16) * ld;
    const float *qp = base + (size_t)head * dh * ld, *kp = base + (size_t)(C + head * dh) * ld, *vp = base + (size_t)(2 * C + head * dh) * ld,
                *dp = base + (size_t)(3 * C + head * 4) * ld;
    const float isq = 1.0f / sqrtf((float)dh);
    float q[16], o[16];
    for (int d = 0; d < dh; ++d) { q[d] = qp[(size_t)d * ld + s] * isq; o[d] = 0.f; }
    float slope = 0.f;
    for (int f = 0; f < 4; ++f) slope += (float)(f + 1) * (sigmoid_f(dp[(size_t)f * ld + s]) * 0.5f);
    slope *= 0.5f;
    float mrun = -INFINITY, lrun = 0.f;
    for (int t = 0; t < T; ++t) {
        float sc = 0.f;
        for (int d = 0; d < dh; ++d) sc = fmaf(kp[(size_t)d * ld + t], q[d], sc);
        sc -= fabsf((float)(t - s)) * slope;
        if (t == s) sc = -100.f;
        const float mnew = fmaxf(mrun, sc), alpha = __expf(mrun - mnew), p = __expf(sc - mnew);
        lrun = lrun * alpha + p;
        for (int d = 0; d < dh; ++d) o[d] = o[d] * alpha + p * vp[(size_t)d * ld + t];
        mrun = mnew;
    }
    const float inv = 1.0f / lrun;
    for (int d = 0; d < dh; ++d) out[((size_t)b * C + head * dh + d) * ld_o + s] = o[d] * inv;
}

int launch_local_attn(const float *qkc, int B, int C, int T, int ld, float *out, int ld_o, hipStream_t st) {
    MI_REQUIRE(ld % 4 == 0 && ld >= T && ((uintptr_t)qkc & 15) == 0, "local_attn: row pitch %d must be a multiple of 4 (T = %d)", ld, T);
    if (C != 192 && C != 384) {
        MI_REQUIRE(C % 4 == 0 && C / 4 <= 16, "local_attn: %d channels not instantiated", C);
        hipLaunchKernelGGL(local_attn_small_kernel, dim3(ceil_div(T, 256), 4, B), dim3(256), 0, st, qkc, C, T, ld, out, ld_o);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    const dim3 grid(ceil_div(T, 64), 4, B);
    if (C == 192) hipLaunchKernelGGL(local_attn_kernel<48>, grid, dim3(128), 0, st, qkc, C, T, ld, out, ld_o);
    else if (C == 384) hipLaunchKernelGGL(local_attn_kernel<96>, grid, dim3(128), 0, st, qkc, C, T, ld, out, ld_o);
    else return set_error(MI_EINVAL, "local_attn: %d channels not instantiated", C);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
