// Multi-head attention core  O = softmax(Q^T K / 8) V  on channel-first tensors, fp32 MFMA.
//
// Stands in for the attention inside nn.MultiheadAttention as called by the reference
// (demucs/transformer.py:418-419,506; 8 heads x 64, no mask, eval mode).
//
// Layout: q[b][h*64 + d][tq], k/v[b][h*64 + d][tk] (tokens contiguous), o like q.
// Workgroup = 4 waves x 32 queries.  Per 32-key sub-tile a wave computes the TRANSPOSED score
// tile S^T[key][query] = K^T Q (keys on MFMA rows, queries on lanes), so that after the online
// softmax the 16 accumulator registers of a lane ARE the B operand of the second product
// O^T[d][query] += V[d][key] P^T[key][query]: no LDS round trip, no cross-lane movement; the
// k order of that product is permuted to the accumulator row map (key = (s&3) + 8(s>>2) + 4h).
#include "common.h"
#include "kernels.h"

namespace mi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HD = 64;       // head dim
constexpr int KT = 64;       // keys per LDS tile
constexpr int VLD = KT + 1;  // V tile row stride (A-operand reads walk rows: odd stride = no bank conflicts)

// Three workgroups per CU (round 4): without the bound hipcc took 192 VGPRs = two waves per SIMD; at 168 the five values it spills are
// written before the key loop and read after it (ISA checked), and the third wave per SIMD hides more of each wave's
// MFMA -> softmax -> MFMA chain: 119.5 -> 123.5 TFLOP/s, fp32 step 102.1 -> 101.0 ms (alternating runs).  Four (128 VGPRs) spills 70
// values inside the loop.
__global__ __launch_bounds__(256, 3) void attention_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                        const float *__restrict__ v, float *__restrict__ o, int Tq, int Tk,
                                                        int64_t q_bs, int64_t kv_bs, int64_t o_bs, int planes, int heads) {
    __shared__ float Ks[HD][KT];
    __shared__ float Vs[HD][VLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware order (workgroups id, id + 8, ... share one of the 8 L2s): all query blocks of one (item, head) plane run on the
    // SAME XCD, so that plane's K / V (1.4 MB at 2 688 keys) is fetched into one L2 instead of eight
    const int nqb = (Tq + 127) / 128;
    const int xj = blockIdx.x >> 3;
    const int pl = (xj / nqb) * 8 + (blockIdx.x & 7);
    if (pl >= planes) return;                  // grid padding (whole workgroup, before any barrier)
    const int head = pl % heads, b = pl / heads;
    const int q0 = (xj % nqb) * 128 + wave * 32;
    const float *qp = q + (size_t)b * q_bs + (size_t)head * HD * Tq;
    const float *kp = k + (size_t)b * kv_bs + (size_t)head * HD * Tk;
    const float *vp = v + (size_t)b * kv_bs + (size_t)head * HD * Tk;

    // Q fragment (B operand of S^T = K^T Q): lane (query li, half lh) holds Q[d = 2s + lh][q0 + li] / 8
    const int qi = q0 + li;
    const bool qok = qi < Tq;
    float qreg[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) qreg[s] = qok ? qp[(size_t)(2 * s + lh) * Tq + qi] * 0.125f : 0.f;

    f32x16 oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float mrun = -INFINITY, lrun = 0.f;     // running max (both halves agree) and this half's partial sum

    // global -> register staging of the next K/V tile (64 x 64 each): thread owns 4 float4 of each
    const int sr = tid >> 4, sc4 = (tid & 15) * 4;            // rows sr + 16*it, columns sc4..sc4+3
    float4 kst[4], vst[4];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = sr + 16 * it;
            if (k0 + sc4 + 3 < Tk) {
                kst[it] = *reinterpret_cast<const float4 *>(kp + (size_t)r * Tk + k0 + sc4);
                vst[it] = *reinterpret_cast<const float4 *>(vp + (size_t)r * Tk + k0 + sc4);
            } else {
                float tk[4] = {0, 0, 0, 0}, tv[4] = {0, 0, 0, 0};
                for (int e = 0; e < 4; ++e)
                    if (k0 + sc4 + e < Tk) { tk[e] = kp[(size_t)r * Tk + k0 + sc4 + e]; tv[e] = vp[(size_t)r * Tk + k0 + sc4 + e]; }
                kst[it] = make_float4(tk[0], tk[1], tk[2], tk[3]); vst[it] = make_float4(tv[0], tv[1], tv[2], tv[3]);
            }
        }
    };
    stage_load(0);
    for (int k0 = 0; k0 < Tk; k0 += KT) {
        __syncthreads();                     // previous tile fully consumed
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = sr + 16 * it;
            *reinterpret_cast<float4 *>(&Ks[r][sc4]) = kst[it];
            Vs[r][sc4] = vst[it].x; Vs[r][sc4 + 1] = vst[it].y; Vs[r][sc4 + 2] = vst[it].z; Vs[r][sc4 + 3] = vst[it].w;
        }
        __syncthreads();
        if (k0 + KT < Tk) stage_load(k0 + KT);   // in flight under this tile's 128 MFMAs
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kb = sub * 32;
            // S^T[key][query]: A = K^T (lane: key li, k = d = 2s + lh), B = Q; fragments prefetched 4 deep
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            float kf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = Ks[2 * s + lh][kb + li];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float a = kf[s & 3];
                if (s + 4 < 32) kf[s & 3] = Ks[2 * (s + 4) + lh][kb + li];
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[s], sacc, 0, 0, 0);
            }
            // first V fragments can already be fetched while the softmax runs
            float vf0[4], vf1[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kc = kb + (s & 3) + 8 * (s >> 2) + 4 * lh;
                vf0[s] = Vs[li][kc]; vf1[s] = Vs[32 + li][kc];
            }
            // register r of lane (li, lh) is key kb + (r&3) + 8(r>>2) + 4 lh, query li
            float mloc = -INFINITY;
            if (k0 + kb + 32 > Tk) {             // ragged last tile only (wave-uniform): mask keys past Tk
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb + (r & 3) + 8 * (r >> 2) + 4 * lh >= Tk) sacc[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float mnew = fmaxf(mrun, mloc);
            // exp through v_exp_f32 (2^x): |x| <= ~30 here, so the argument rounding costs <= 2e-6 relative on a
            // probability; masked keys give exp(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(sacc[r] - mnew);
                sacc[r] = p;
                psum += p;
            }
            if (__any(mnew != mrun)) {          // the running max moved for some query of this wave: rescale
                const float alpha = __expf(mrun - mnew);       // exp(-inf) = 0 on the first tile
                lrun *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
                mrun = mnew;
            }
            lrun += psum;
            // O^T[d][query] += V[d][key] P^T[key][query]: A = V (lane: d = dt*32 + li, k = key(s, lh)), B = P regs
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float a0 = vf0[s & 3], a1 = vf1[s & 3];
                if (s + 4 < 16) {
                    const int kc = kb + ((s + 4) & 3) + 8 * ((s + 4) >> 2) + 4 * lh;
                    vf0[s & 3] = Vs[li][kc]; vf1[s & 3] = Vs[32 + li][kc];
                }
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, sacc[s], oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, sacc[s], oacc[1], 0, 0, 0);
            }
        }
    }
    const float ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.0f / ltot;
    if (qok) {
        float *op = o + (size_t)b * o_bs + (size_t)head * HD * Tq + qi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dd = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                op[(size_t)dd * Tq] = oacc[dt][r] * inv;
            }
    }
}

// gemm of the reduced-precision modes: attention_half.hip (built without the forced VGPR-form MFMA)
int launch_attention_half(const float *q, const float *k, const float *v, float *o, int B, int heads, int Tq, int Tk, int64_t q_bs,
                          int64_t kv_bs, int64_t o_bs, int dtype, hipStream_t st, void *oh, int64_t oh_n);

int launch_attention(const float *q, const float *k, const float *v, float *o, int B, int heads, int Tq, int Tk, int64_t q_bs,
                     int64_t kv_bs, int64_t o_bs, int dtype, hipStream_t st, void *oh, int64_t oh_n) {
    MI_REQUIRE(!oh || dtype != MI_DTYPE_F32, "attention: the operand-image output exists in the half modes only");
    MI_REQUIRE(Tk % 4 == 0, "attention: Tk %% 4 != 0 (%d)", Tk);
    MI_REQUIRE(((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && kv_bs % 4 == 0, "attention: k/v must be 16-byte aligned");
    MI_REQUIRE(dtype == MI_DTYPE_F32 || dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16, "attention: dtype %d", dtype);
    if (dtype != MI_DTYPE_F32) return launch_attention_half(q, k, v, o, B, heads, Tq, Tk, q_bs, kv_bs, o_bs, dtype, st, oh, oh_n);
    const int planes = B * heads;
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)(ceil_div(Tq, 128) * ((planes + 7) / 8) * 8)), dim3(256), 0, st, q, k, v, o, Tq, Tk, q_bs,
                       kv_bs, o_bs, planes, heads);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
