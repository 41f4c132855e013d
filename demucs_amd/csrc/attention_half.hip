// Multi-head attention core for the reduced-precision compute modes (see attention.hip for the layout and the
// transposed-score scheme; reference call sites demucs/transformer.py:418-419,506).
#include "common.h"
#include "kernels.h"

namespace mi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HD = 64;       // head dim
constexpr int KT = 64;       // keys per LDS tile

// ---------------------------------------------------------------------------------------------
// Reduced-precision modes (mi_config.dtype = bf16 / fp16): both products on v_mfma_f32_32x32x16_{bf16,f16}.
// Q (pre-scaled by 1/8), K, V and the probabilities P are rounded to the operand type; scores, the online softmax
// (max, exp, sum), the rescaling and the output accumulators are float32.  Same transposed-score scheme as above:
//   S^T[key][query] = sum_d K[d][key] Q[d][query]      A = K^T from LDS [d octet][key][8], B = Q fragments (registers)
//   O^T[d][query]  += sum_key V[d][key] P^T[key][query] B = the softmaxed accumulator registers, packed 8 at a time;
// the contraction order over keys follows the accumulator row map (register r of half h is key (r&3) + 8(r>>2) + 4h),
// and V is written to LDS with its keys in that same order, so an A fragment is one 16-byte word as well.
template <int HT>
__device__ __forceinline__ unsigned apack2(float a, float b) {
    if (HT == MI_DTYPE_BF16) {
        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
        const v2 h = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, h);
    } else {
        typedef _Float16 v2 __attribute__((ext_vector_type(2)));
        const v2 h = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, h);
    }
}
template <int HT>
__device__ __forceinline__ f32x16 amfma16(const uint4 a, const uint4 b, const f32x16 c) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    if (HT == MI_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

constexpr int VROW = KT / 8 + 1;     // V tile row pitch in 16-byte words (odd: conflict-free A-fragment reads down a column of rows)

template <int HT>
__global__ __launch_bounds__(256) void attention_half_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                             const float *__restrict__ v, float *__restrict__ o, int Tq, int Tk,
                                                             int64_t q_bs, int64_t kv_bs, int64_t o_bs, void *__restrict__ oh, int64_t oh_n) {
    __shared__ uint4 Ks[HD / 8][KT];         // [d octet][key] -> 8 consecutive d
    __shared__ uint4 Vs[HD][VROW];           // [d][key octet in accumulator-row order]
    {   // accumulators in AGPRs (see gemm_x6.hip)
        float agpr_anchor = 0.f;
        asm volatile("; accumulators in AGPRs %0" : "+a"(agpr_anchor));
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const float *qp = q + (size_t)b * q_bs + (size_t)head * HD * Tq;
    const float *kp = k + (size_t)b * kv_bs + (size_t)head * HD * Tk;
    const float *vp = v + (size_t)b * kv_bs + (size_t)head * HD * Tk;

    // Q fragments (B operand of S^T): lane (query li, half lh), k step s holds Q[d = 16 s + 8 lh + j][query] / 8, j = 0..7
    const int qi = q0 + li;
    const bool qok = qi < Tq;
    uint4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = qok ? qp[(size_t)(16 * s + 8 * lh + j) * Tq + qi] * 0.125f : 0.f;
        qf[s] = make_uint4(apack2<HT>(t[0], t[1]), apack2<HT>(t[2], t[3]), apack2<HT>(t[4], t[5]), apack2<HT>(t[6], t[7]));
    }

    f32x16 oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float mrun = -INFINITY, lrun = 0.f;

    // staging: K unit u = tid + 256 i -> (octet u >> 6, key u & 63): 8 d values of one key (coalesced along keys);
    //          V unit u -> (d = u >> 3, word w = u & 7): word w = (block w >> 2, t (w >> 1) & 1, half w & 1) holds keys
    //          32 block + 16 t + 4 half + {0..3} and + 8 + {0..3}: two aligned float4
    float kst[2][8];
    float4 vst[2][2];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + 256 * i;
            const int key = k0 + (u & 63), oc = u >> 6;
            const bool ok = key < Tk;
#pragma unroll
            for (int j = 0; j < 8; ++j) kst[i][j] = ok ? kp[(size_t)(8 * oc + j) * Tk + key] : 0.f;
            const int d = u >> 3, w = u & 7;
            const int kb = k0 + 32 * (w >> 2) + 16 * ((w >> 1) & 1) + 4 * (w & 1);
            const float *src = vp + (size_t)d * Tk + kb;
            vst[i][0] = kb + 3 < Tk ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            vst[i][1] = kb + 11 < Tk ? *reinterpret_cast<const float4 *>(src + 8) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    stage_load(0);
    for (int k0 = 0; k0 < Tk; k0 += KT) {
        __syncthreads();                     // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + 256 * i;
            Ks[u >> 6][u & 63] = make_uint4(apack2<HT>(kst[i][0], kst[i][1]), apack2<HT>(kst[i][2], kst[i][3]),
                                            apack2<HT>(kst[i][4], kst[i][5]), apack2<HT>(kst[i][6], kst[i][7]));
            Vs[u >> 3][u & 7] = make_uint4(apack2<HT>(vst[i][0].x, vst[i][0].y), apack2<HT>(vst[i][0].z, vst[i][0].w),
                                           apack2<HT>(vst[i][1].x, vst[i][1].y), apack2<HT>(vst[i][1].z, vst[i][1].w));
        }
        __syncthreads();
        if (k0 + KT < Tk) stage_load(k0 + KT);   // in flight under this tile's MFMAs and softmax
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kb = sub * 32;
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc = amfma16<HT>(Ks[2 * s + lh][kb + li], qf[s], sacc);
            // V fragments for the second product can be fetched while the softmax runs
            uint4 vf[2][2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int t = 0; t < 2; ++t) vf[dt][t] = Vs[dt * 32 + li][4 * sub + 2 * t + lh];
            // register r of lane (li, lh) is key kb + (r&3) + 8(r>>2) + 4 lh, query li
            if (k0 + kb + 32 > Tk) {             // ragged last tile only (wave-uniform): mask keys past Tk
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb + (r & 3) + 8 * (r >> 2) + 4 * lh >= Tk) sacc[r] = -INFINITY;
            }
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
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
                const float alpha = __expf(mrun - mnew);
                lrun *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
                mrun = mnew;
            }
            lrun += psum;
            uint4 pb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                pb[t] = make_uint4(apack2<HT>(sacc[8 * t], sacc[8 * t + 1]), apack2<HT>(sacc[8 * t + 2], sacc[8 * t + 3]),
                                   apack2<HT>(sacc[8 * t + 4], sacc[8 * t + 5]), apack2<HT>(sacc[8 * t + 6], sacc[8 * t + 7]));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                oacc[0] = amfma16<HT>(vf[0][t], pb[t], oacc[0]);
                oacc[1] = amfma16<HT>(vf[1][t], pb[t], oacc[1]);
            }
        }
    }
    const float ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.0f / ltot;
    if (qok && oh) {
        // the output only feeds out_proj's matrix product: written as its 16-bit operand image [channel / 8][b * Tq + query][8]
        // (gemm_half.hip: conv_gemm_half_img_kernel); four consecutive channels of a lane = one 8-byte store
        uint2 *img = reinterpret_cast<uint2 *>(oh);
        const size_t n = (size_t)b * Tq + qi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int oct = (head * HD + dt * 32 + 8 * rq) >> 3;
                img[((size_t)oct * oh_n + n) * 2 + lh] =
                    make_uint2(pack_half2(HT, oacc[dt][4 * rq] * inv, oacc[dt][4 * rq + 1] * inv),
                               pack_half2(HT, oacc[dt][4 * rq + 2] * inv, oacc[dt][4 * rq + 3] * inv));
            }
    } else if (qok) {
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

int launch_attention_half(const float *q, const float *k, const float *v, float *o, int B, int heads, int Tq, int Tk, int64_t q_bs,
                          int64_t kv_bs, int64_t o_bs, int dtype, hipStream_t st, void *oh, int64_t oh_n) {
    MI_REQUIRE(!oh || (oh_n >= (int64_t)B * Tq && ((uintptr_t)oh & 15) == 0), "attention: output image too small or misaligned");
    const dim3 grid(ceil_div(Tq, 128), heads, B);
    if (dtype == MI_DTYPE_BF16)
        hipLaunchKernelGGL(attention_half_kernel<MI_DTYPE_BF16>, grid, dim3(256), 0, st, q, k, v, o, Tq, Tk, q_bs, kv_bs, o_bs, oh, oh_n);
    else
        hipLaunchKernelGGL(attention_half_kernel<MI_DTYPE_F16>, grid, dim3(256), 0, st, q, k, v, o, Tq, Tk, q_bs, kv_bs, o_bs, oh, oh_n);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
