// Multi-head attention core of the reduced-precision compute modes on 16-bit PER-HEAD operand tensors
// (reference call sites demucs/transformer.py:339-377,466-512: nn.MultiheadAttention, 8 heads x 64).
//
// Q, K and V are consumed by nothing but this kernel, so the projections' epilogues (MI_FLAG_HEADS, gemm_tile.h) write
// them ONLY as bf16 / fp16, token-major per head:   X[b][head][token][64]   (128-byte rows, no float32 copy).
//   * K / V tiles (64 keys = 8 KiB each) go global -> LDS by DMA (global_load_lds_dwordx4) through a 3-stage ring behind a
//     counted s_waitcnt vmcnt: two tiles in flight under the products of the current one, ONE s_barrier per tile, no
//     staging registers and no conversion in the loop.  LDS is lane-linear for the DMA; the bank swizzle is applied to the
//     SOURCE address: LDS row `key`, 16-byte slot c' holds chunk c' ^ f(key), f(key) = ((key >> 1) & 1) << 2 | (key >> 2) & 3,
//     which makes both the row reads of K (ds_read_b128: S^T = K^T Q needs 8 consecutive d of one key) and the transposed
//     reads of V (ds_read_b64_tr_b16: O^T = V^T-as-A . P^T needs 4 consecutive keys of one d) conflict-free.
//   * a wave owns 64 queries (two 32-query blocks) and re-uses every K / V fragment for both: 16 B of LDS reads per MFMA
//     cycle per wave, half of what 32-query waves need (which is the LDS limit at two workgroups per CU).
//   * same transposed-score scheme as attention.hip: the softmaxed accumulator registers ARE the B operand of the second
//     product; scores, online softmax (exp2 domain: p = 2^(s c - m c), c = log2(e) / 8), rescaling and output accumulators
//     are float32; the output is written as out_proj's operand image (or float32 for the kernel-level test).
#include "common.h"
#include "kernels.h"

namespace mi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void agvoid_t;
typedef __attribute__((address_space(3))) void alvoid_t;

namespace {
constexpr int AKT = 64;                  // keys per tile
constexpr int AROW = 8;                  // 16-byte words per token row (64 x 16 bit)
constexpr int ASTAGE = 2 * AKT * AROW;   // words per ring stage: K tile then V tile (16 KiB)

template <int HT>
__device__ __forceinline__ unsigned hpack2(float a, float b) {
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
__device__ __forceinline__ f32x16 hmfma(const uint4 a, const uint4 b, const f32x16 c) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    if (HT == MI_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// value of the lane 32 away combined with this lane's: v_permlane32_swap (VALU) instead of the ds_bpermute that __shfl_xor(x, 32)
// becomes (an LDS round trip and an lgkmcnt(0) wait in the middle of the softmax)
__device__ __forceinline__ float other_half(float x) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
// max of three without the canonicalising v_max hipcc puts in front of fmaxf on matrix-pipe results (the softmax is bound by
// VALU issue slots: ~4.6 cycles of the SIMD per wave instruction, counters in tools/micro/README.md)
__device__ __forceinline__ float max3(float a, float b, float cc) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(cc));
    return r;
}
// max(a, b) over this lane and the lane 32 away
__device__ __forceinline__ float other_half_max(float a, float b) {
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return other_half(m);
}
__device__ __forceinline__ int swz(int key) { return (((key >> 1) & 1) << 2) | ((key >> 2) & 3); }
}  // namespace

template <int HT>
__global__ __launch_bounds__(256, 1) void attention_heads_kernel(const uint4 *__restrict__ qimg, const uint4 *__restrict__ kimg,
                                                                 const uint4 *__restrict__ vimg, const uint4 *__restrict__ zero, int planes,
                                                                 int heads, int Tq, int Tk, int Tqp, int Tkp, void *__restrict__ oh, int64_t oh_n,
                                                                 float *__restrict__ o, int64_t o_bs) {
    __shared__ __attribute__((aligned(16))) uint4 smem[3 * ASTAGE];
#ifdef MI_AGPR_ACC          // `make MFMA_FORM=agpr`: accumulators in AGPRs (see gemm_x6.hip); the default build keeps this kernel in the
    {                       // VGPR form: at two waves per SIMD hipcc splits the 256 registers 128 / 128 otherwise and spills the Q fragments
        float agpr_anchor = 0.f;
        asm volatile("; accumulators in AGPRs %0" : "+a"(agpr_anchor));
    }
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware order (workgroups id, id + 8, ... share one of the 8 L2s): all query blocks of one (item, head) plane run on the
    // SAME XCD, so its K / V (688 KiB at 2 688 keys) is fetched into one L2 instead of eight
    const int nqb = (Tq + 255) / 256;
    const int xj = blockIdx.x >> 3;
    const int pl = (xj / nqb) * 8 + (blockIdx.x & 7);
    if (pl >= planes) return;                                      // grid padding (whole workgroup, before any barrier)
    const int head = pl % heads, b = pl / heads;
    const size_t plane = (size_t)pl;
    const int q0 = (xj % nqb) * 256 + wave * 64;
    const bool active = q0 < Tq;                                   // wave-uniform: a wave past the last query only moves tiles
    const uint4 *kp = kimg + plane * (size_t)Tkp * AROW, *vp = vimg + plane * (size_t)Tkp * AROW;
    const int nt = (Tk + AKT - 1) / AKT;

    // ---- DMA: this wave moves keys [16 wave, 16 wave + 16) of the K tile and of the V tile: 2 + 2 wave instructions of 1 KiB;
    //      lane -> (key 8 j + (lane >> 3), slot lane & 7) holds source chunk slot ^ f(key)
    const int dkey = 16 * wave + (lane >> 3);
    const int dch0 = (lane & 7) ^ swz(dkey), dch1 = (lane & 7) ^ swz(dkey + 8);
    // full tiles: wave-uniform tile base (scalar arithmetic) + two constant per-lane byte offsets -- no VALU slot per tile; the
    // ragged last tile (rows past Tk come from the zero page) takes per-lane addresses
    const unsigned doff0 = (unsigned)((dkey * AROW + dch0) * 16), doff1 = (unsigned)(((dkey + 8) * AROW + dch1) * 16);
#define MI_ATT_TILE(t, stage)                                                                                          \
    do {                                                                                                               \
        uint4 *sk = smem + (stage) * ASTAGE + 16 * wave * AROW, *sv = sk + AKT * AROW;                                 \
        if (((t) + 1) * AKT <= Tk) {                                                                                   \
            const uint4 *kt_ = kp + (size_t)(t) * AKT * AROW, *vt_ = vp + (size_t)(t) * AKT * AROW;                    \
            lds_dma16_s(kt_, doff0, sk);                                                                               \
            lds_dma16_s(kt_, doff1, sk + 8 * AROW);                                                                    \
            lds_dma16_s(vt_, doff0, sv);                                                                               \
            lds_dma16_s(vt_, doff1, sv + 8 * AROW);                                                                    \
        } else {                                                                                                       \
            const int k0_ = (t) * AKT + dkey;                                                                          \
            const uint4 *g0 = (k0_ < Tk) ? kp + (size_t)k0_ * AROW + dch0 : zero;                                      \
            const uint4 *g1 = (k0_ + 8 < Tk) ? kp + (size_t)(k0_ + 8) * AROW + dch1 : zero;                            \
            const uint4 *h0 = (k0_ < Tk) ? vp + (size_t)k0_ * AROW + dch0 : zero;                                      \
            const uint4 *h1 = (k0_ + 8 < Tk) ? vp + (size_t)(k0_ + 8) * AROW + dch1 : zero;                            \
            lds_dma16(g0, sk);                                                                                         \
            lds_dma16(g1, sk + 8 * AROW);                                                                              \
            lds_dma16(h0, sv);                                                                                         \
            lds_dma16(h1, sv + 8 * AROW);                                                                              \
        }                                                                                                              \
    } while (0)

    MI_ATT_TILE(0, 0);
    if (nt > 1) MI_ATT_TILE(1, 1);

    // ---- Q fragments (B operand of S^T): lane (query li, half lh), k step s = chunk 2 s + lh of the query's row
    uint4 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + 32 * qb + li;
        const uint4 *qrow = qimg + (plane * (size_t)Tqp + (qi < Tq ? qi : 0)) * AROW;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qb][s] = qrow[2 * s + lh];
    }
    // everything issued so far (two tiles, the Q rows) has landed before the loop: inside it only DMA is ever in flight.  The
    // empty asm statements USE the Q registers, so hipcc places its own wait for those loads here; without them it carries
    // "Q may still be in flight" into the loop and its vmcnt(7) ... vmcnt(0) in front of the first products drain the ring
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[qb][s].x), "+v"(qf[qb][s].y), "+v"(qf[qb][s].z), "+v"(qf[qb][s].w));

    f32x16 oacc[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][dt][r] = 0.f;
    float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
    const float c = 0.125f * 1.44269504088896341f;                 // scores / sqrt(64), in the exp2 domain

    // ---- LDS read addresses (16-byte word index for K, byte offset for V), lane parts
    //  K fragment (sub, s): row 32 sub + li, chunk (2 s + lh) ^ f(li)
    int koff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koff[s] = li * AROW + ((2 * s + lh) ^ swz(li));
    //  V transposed read (sub, t, e, dt): 16-lane group g = lane >> 4 reads keys 32 sub + 16 t + 8 e + 4 lh + {0..3} x columns
    //  32 dt + 16 (g & 1) + {0..15}; lane 4 q + p of the group supplies row q, columns 4 p .. 4 p + 3
    const int vq = (lane & 15) >> 2, vpp = lane & 3, vg = (lane >> 4) & 1;
    int voff[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int key = 8 * e + 4 * lh + vq;
            voff[e][dt] = key * 128 + (((4 * dt + 2 * vg + (vpp >> 1)) ^ swz(key)) * 16) + 8 * (vpp & 1);
        }

    // S^T of one 32-key block for both query blocks: K fragments (shared), 8 products.  Keys past Tk (ragged last tile only,
    // wave-uniform test) enter as -inf through the accumulator's initial value, so no select touches the scores afterwards.
    auto score = [&](const uint4 *Ks, int sub, int kb, f32x16(&sc)[2]) {
        uint4 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = Ks[32 * sub * AROW + koff[s]];
        if (kb + 32 > Tk) {          // ragged last tile: a real branch (the asm keeps hipcc from turning it into 32 selects per call)
            asm volatile("; ragged key block");
            f32x16 init;
#pragma unroll
            for (int r = 0; r < 16; ++r) init[r] = (kb + (r & 3) + 8 * (r >> 2) + 4 * lh >= Tk) ? -INFINITY : 0.f;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                sc[qb] = hmfma<HT>(kf[0], qf[qb][0], init);
#pragma unroll
                for (int s = 1; s < 4; ++s) sc[qb] = hmfma<HT>(kf[s], qf[qb][s], sc[qb]);
            }
        } else {
            f32x16 zero16;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                sc[qb] = hmfma<HT>(kf[0], qf[qb][0], zero16);
#pragma unroll
                for (int s = 1; s < 4; ++s) sc[qb] = hmfma<HT>(kf[s], qf[qb][s], sc[qb]);
            }
        }
    };
    // online softmax of one 32-key block's scores and the second product, both query blocks
    auto finish = [&](const char *Vs, int sub, f32x16(&sc)[2]) {
        // V fragments (A operand): element j of lane half lh is key 16 tt + 8 (j >> 2) + 4 lh + (j & 3), the key order of the
        // accumulator rows that become the B operand
        uint4 vf[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const char *base = Vs + (32 * sub + 16 * tt) * 128;
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)(base + voff[0][dt]));
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)(base + voff[1][dt]));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                vf[dt][tt] = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // register r of lane (li, lh) is key (r & 3) + 8 (r >> 2) + 4 lh of the block, query li
            float mloc = max3(max3(max3(sc[qb][0], sc[qb][1], sc[qb][2]), sc[qb][3], sc[qb][4]), max3(sc[qb][5], sc[qb][6], sc[qb][7]),
                              max3(max3(sc[qb][8], sc[qb][9], sc[qb][10]), sc[qb][11], sc[qb][12]));
            mloc = max3(mloc, max3(sc[qb][13], sc[qb][14], sc[qb][15]), other_half_max(mloc, max3(sc[qb][13], sc[qb][14], sc[qb][15])));
            const float mnew = fmaxf(mrun[qb], mloc);
            if (__any(mnew != mrun[qb])) {
                const float alpha = __builtin_amdgcn_exp2f((mrun[qb] - mnew) * c);
                lrun[qb] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[qb][0][r] *= alpha; oacc[qb][1][r] *= alpha; }
                mrun[qb] = mnew;
            }
            // p = 2^(s c - m c): the multiply-adds and the row sum as PACKED float32 operations (two values per VALU slot)
            const v2f cc2 = splat2(c), nm2 = splat2(-mnew * c);
            v2f ps2 = splat2(0.f);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f a = fma2((v2f){sc[qb][r], sc[qb][r + 1]}, cc2, nm2);
                const v2f pp = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                sc[qb][r] = pp.x; sc[qb][r + 1] = pp.y;
                ps2 += pp;
            }
            lrun[qb] += ps2.x + ps2.y;
            uint4 pb[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                pb[tt] = make_uint4(hpack2<HT>(sc[qb][8 * tt], sc[qb][8 * tt + 1]), hpack2<HT>(sc[qb][8 * tt + 2], sc[qb][8 * tt + 3]),
                                    hpack2<HT>(sc[qb][8 * tt + 4], sc[qb][8 * tt + 5]), hpack2<HT>(sc[qb][8 * tt + 6], sc[qb][8 * tt + 7]));
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                oacc[qb][0] = hmfma<HT>(vf[0][tt], pb[tt], oacc[qb][0]);
                oacc[qb][1] = hmfma<HT>(vf[1][tt], pb[tt], oacc[qb][1]);
            }
        }
    };

    // Software pipeline over 32-key blocks: the scores of block i + 1 are issued to the matrix pipe BEFORE the softmax of block
    // i runs on the VALU (two named accumulator sets), so one wave keeps both pipes busy; the only barrier of an iteration is
    // at its end, behind a full drain of the DMA queue: tile t + 2, issued at the top of iteration t, is first read in
    // iteration t + 1 (block 0 of tile t + 2 is scored there), a whole tile time after its issue.
    __builtin_amdgcn_s_barrier();                 // tiles 0 and 1 have landed for every wave (each waited for its own parts above)
    f32x16 sA[2], sB[2];
    if (active) score(smem, 0, 0, sA);
    int stage = 0;
    for (int t = 0; t < nt; ++t) {
        const int nstage = stage == 2 ? 0 : stage + 1;
        // stage (t + 2) % 3 held tile t - 1: last read in iteration t - 1, which every wave has left through its barrier
        if (t + 2 < nt) MI_ATT_TILE(t + 2, stage == 0 ? 2 : stage - 1);
        if (active) {
            const uint4 *Ks = smem + stage * ASTAGE;
            const char *Vs = reinterpret_cast<const char *>(Ks + AKT * AROW);
            score(Ks, 1, t * AKT + 32, sB);
            finish(Vs, 0, sA);
            if (t + 1 < nt) score(smem + nstage * ASTAGE, 0, (t + 1) * AKT, sA);
            finish(Vs, 1, sB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage = nstage;
    }
#undef MI_ATT_TILE
    if (!active) return;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + 32 * qb + li;
        const float ltot = lrun[qb] + other_half(lrun[qb]);
        const float inv = 1.0f / ltot;
        if (qi >= Tq) continue;
        if (oh) {
            // out_proj's 16-bit operand image [channel / 8][b * Tq + query][8] (gemm_half.hip): four consecutive channels of a
            // lane = one 8-byte store
            uint2 *img = reinterpret_cast<uint2 *>(oh);
            const size_t n = (size_t)b * Tq + qi;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int oct = (head * 64 + dt * 32 + 8 * rq) >> 3;
                    img[((size_t)oct * oh_n + n) * 2 + lh] =
                        make_uint2(pack_half2(HT, oacc[qb][dt][4 * rq] * inv, oacc[qb][dt][4 * rq + 1] * inv),
                                   pack_half2(HT, oacc[qb][dt][4 * rq + 2] * inv, oacc[qb][dt][4 * rq + 3] * inv));
                }
        } else {
            float *op = o + (size_t)b * o_bs + (size_t)head * 64 * Tq + qi;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) op[(size_t)(dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * Tq] = oacc[qb][dt][r] * inv;
        }
    }
}

// q / k / v: per-head token-major 16-bit tensors [B][heads][T pitch][64]; the pitches are multiples of 64 rows so that every
// tile read stays inside the allocation (rows past Tk are never used: their probabilities are exactly zero and the DMA reads
// the zero page instead).  Output: `oh` (operand image with oh_n columns) or float32 o (B, heads * 64, Tq).
int launch_attention_heads(const void *q, const void *k, const void *v, const void *zero_page, int B, int heads, int Tq, int Tk, int Tq_pitch,
                           int Tk_pitch, int dtype, void *oh, int64_t oh_n, float *o, int64_t o_bs, hipStream_t st) {
    MI_REQUIRE(dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16, "attention: per-head operands exist in the half modes only (dtype %d)", dtype);
    MI_REQUIRE(Tq_pitch >= Tq && Tk_pitch >= Tk, "attention: row pitch below the token count");
    MI_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)zero_page) & 15) == 0, "attention: operand tensors must be 16-byte aligned");
    MI_REQUIRE((oh != nullptr) != (o != nullptr), "attention: exactly one output form");
    MI_REQUIRE(!oh || (oh_n >= (int64_t)B * Tq && ((uintptr_t)oh & 15) == 0), "attention: output image too small or misaligned");
    const int planes = B * heads;
    const dim3 grid((unsigned)(ceil_div(Tq, 256) * ((planes + 7) / 8) * 8));
    const uint4 *q4 = (const uint4 *)q, *k4 = (const uint4 *)k, *v4 = (const uint4 *)v, *z4 = (const uint4 *)zero_page;
    if (dtype == MI_DTYPE_BF16)
        hipLaunchKernelGGL(attention_heads_kernel<MI_DTYPE_BF16>, grid, dim3(256), 0, st, q4, k4, v4, z4, planes, heads, Tq, Tk, Tq_pitch, Tk_pitch, oh, oh_n, o, o_bs);
    else
        hipLaunchKernelGGL(attention_heads_kernel<MI_DTYPE_F16>, grid, dim3(256), 0, st, q4, k4, v4, z4, planes, heads, Tq, Tk, Tq_pitch, Tk_pitch, oh, oh_n, o, o_bs);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
