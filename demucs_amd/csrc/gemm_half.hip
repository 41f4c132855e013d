// Implicit-GEMM convolution / linear layer with bf16 or fp16 OPERANDS on the 16x matrix pipe
// (v_mfma_f32_32x32x16_bf16 / _f16), fp32 accumulation: the engine's reduced-precision compute modes
// (mi_config.dtype = MI_DTYPE_BF16 / MI_DTYPE_F16; BASELINE.json configs[2], configs[4]).
//
// Same contract as gemm_conv.hip (mi_conv_desc: table-driven gather, fused epilogues of gemm_tile.h); what changes:
//   * weights are converted ONCE at load time into the LDS tile image  Wh[K/8][Mpad][8 x 2 bytes]  (k-octet major), so
//     a 32x32x16 A fragment (lane l: row l & 31, k = 8 (l >> 5) .. +7) is one 16-byte word and a K step's tile is four
//     contiguous BM x 16 B runs;
//   * activations stay float32 in HBM (every norm / statistic / residual downstream reads them in float32); thread
//     (column n, k half) loads its 16 k values of the K step (coalesced along n, through the same gather table),
//     rounds them to the operand type (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round-to-nearest-even) and writes two
//     16-byte words of the B tile image  Bs[k octet][BN][8];
//   * K step 32 = two MFMA k steps; LDS double buffer, global -> register prefetch of step k+1 under the MFMAs of k.
// Bias, GroupNorm / LayerNorm statistics, GELU / GLU / sigmoid, LayerScale and residual adds run in float32 on the
// float32 accumulators exactly as in the fp32 mode.
#include "gemm_tile.h"

namespace mi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int HK = 32;      // K step of the half-precision main loop

template <int HT>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    if (HT == MI_DTYPE_BF16) {
        const bf16x2 h = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, h);
    } else {
        const f16x2 h = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, h);
    }
}

template <int HT>
__device__ __forceinline__ f32x16 mfma16(const uint4 a, const uint4 b, const f32x16 c) {
    if (HT == MI_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Wt[Kpad][Mpad] fp32 -> Wh[Kh/8][Mpad][8] (Kh = Kpad rounded up to 32; the tail is zero)
template <int HT>
__global__ void pack_half_kernel(const float *__restrict__ wt, int Kpad, int Kh, int Mpad, unsigned short *__restrict__ wh) {
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= (size_t)Kh * Mpad) return;
    const int k = (int)(idx / Mpad), m = (int)(idx % Mpad);
    const float x = k < Kpad ? wt[idx] : 0.f;
    const unsigned p = pack2<HT>(x, 0.f);
    wh[((size_t)(k >> 3) * Mpad + m) * 8 + (k & 7)] = (unsigned short)(p & 0xffffu);
}

int launch_pack_half(const float *wt, int Kpad, int Mpad, int dtype, void *wh, hipStream_t st) {
    MI_REQUIRE(dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16, "pack_half: dtype %d", dtype);
    const int Kh = (Kpad + HK - 1) / HK * HK;
    const size_t n = (size_t)Kh * Mpad;
    if (dtype == MI_DTYPE_BF16)
        hipLaunchKernelGGL(pack_half_kernel<MI_DTYPE_BF16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wt, Kpad, Kh, Mpad, (unsigned short *)wh);
    else
        hipLaunchKernelGGL(pack_half_kernel<MI_DTYPE_F16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wt, Kpad, Kh, Mpad, (unsigned short *)wh);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int HT, int WM, int WN, int TM, int TN, int EPI, int LFLAGS, bool PLAIN>
// tiles of up to 96 rows: three workgroups per CU (their accumulators leave room: <= 168 registers with nothing in scratch; the 128- and
// 256-row variants would spill up to 557 values at that bound)
__global__ __launch_bounds__(256, (WM * TM * TN <= 3 ? 3 : 2)) void conv_gemm_half_kernel(const mi_conv_desc d, const int N, const int MT, const int Gm) {
    constexpr int BM = WM * TM * 32;
    static_assert(WN * TN * 32 == BN, "block N tile is 128");
    static_assert(WM * WN == 4, "4 waves");
    __shared__ uint4 As[2][4][BM];
    __shared__ uint4 Bs[2][4][BN];
    {   // accumulators in AGPRs (see gemm_x6.hip: VGPR-form bf16 MFMA streams disturbed co-resident processes on this pool)
        float agpr_anchor = 0.f;
        asm volatile("; accumulators in AGPRs %0" : "+a"(agpr_anchor));
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    if (!tile_of_block(MT, Gm, N, mt, nt)) return;      // grid padding (whole workgroup, before any barrier)
    const int m0 = mt * BM, n0 = nt * BN;
    const int P = d.O1 * d.O2;
    const int o2v = d.o2_valid ? d.o2_valid : d.O2;
    const int nk = (d.Kpad + HK - 1) / HK;

    // ---- A loader: the K step's image is 4 runs of BM 16-byte words ----------------------------------
    constexpr int A_W = 4 * BM, A_FULL = A_W / 256, A_REM = A_W % 256, A_SLOTS = A_FULL + (A_REM ? 1 : 0);
    static_assert(A_SLOTS <= 4, "A tile fits four 16-byte words per thread");
    const bool a_on0 = tid < A_W, a_on1 = tid + 256 < A_W, a_on2 = tid + 512 < A_W, a_on3 = tid + 768 < A_W;
    const int ai0 = a_on0 ? tid : 0, ai1 = a_on1 ? tid + 256 : 0, ai2 = a_on2 ? tid + 512 : 0, ai3 = a_on3 ? tid + 768 : 0;
    const int ao0 = ai0 / BM, am0 = ai0 % BM, ao1 = ai1 / BM, am1 = ai1 % BM, ao2 = ai2 / BM, am2 = ai2 % BM, ao3 = ai3 / BM, am3 = ai3 % BM;
    const uint4 *wh = reinterpret_cast<const uint4 *>(d.wh);
    const uint4 *ap0 = wh + (size_t)ao0 * d.Mpad + m0 + am0;
    const uint4 *ap1 = wh + (size_t)ao1 * d.Mpad + m0 + am1;
    const uint4 *ap2 = wh + (size_t)ao2 * d.Mpad + m0 + am2;
    const uint4 *ap3 = wh + (size_t)ao3 * d.Mpad + m0 + am3;
    const size_t a_step = (size_t)4 * d.Mpad;

    // ---- B loader: thread = (column bn, k half bh): k = 16 bh .. 16 bh + 15 of the K step ---------------
    const int bn = tid & 127;
    const int bh = __builtin_amdgcn_readfirstlane(tid >> 7);
    const ColInfo lc = decompose(n0 + bn, N, P, d.O2, PLAIN ? d.O2 : o2v);
    const int i1b = lc.o1 * d.S1, i2b = lc.o2 * d.S2;
    const float *xcol = d.x + (size_t)lc.b * d.x_bstride + (PLAIN ? (size_t)lc.p : (size_t)i1b * (d.x_ld ? d.x_ld : d.D2) + i2b);

    // TWO register sets: the loads of K steps k+1 and k+2 are both in flight under the MFMAs of step k (the main loop is
    // latency bound otherwise: one 24 KiB tile per workgroup in flight keeps the L2 at ~10 TB/s and the matrix pipe at 11 %)
    uint4 areg0 = make_uint4(0, 0, 0, 0), areg1 = areg0, areg2 = areg0, areg3 = areg0;      // weights (L2 resident) and gathered activations: one step ahead
    float breg[1][16];
    // the gather table through the constant address space: the entries of a K step are wave-uniform, and only loads the
    // compiler can prove invariant become scalar (s_load) -- as per-lane loads they put a second dependent memory round
    // trip in front of every activation load
    typedef int ktab_i4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) ktab_i4 *ktab_ptr;
    static_assert(sizeof(mi_ktab_entry) == 16, "gather table entry = 4 dwords (off, d1, d2, ci)");
    const ktab_ptr ktab_c = (ktab_ptr)(uintptr_t)d.ktab;
    // PLAIN (1x1 / linear, channel stride P, P % 4 == 0): thread = (column quad q, pair group pg) loads rows 2 pg, 2 pg + 1,
    // 16 + 2 pg, 17 + 2 pg of the K step as float4 along n (4 VMEM instructions instead of 16 dword loads), packs the two
    // rows of a pair per column and writes ONE 16-byte word per pair into the pair-interleaved image
    //     Bq[k octet][pair in octet][BN] (one dword = 2 operand values per column),
    // from which a fragment is four conflict-free ds_read_b32 (consecutive lanes = consecutive columns).
    const int pq = tid & 31, pg = tid >> 5;
    const ColInfo pc = decompose(n0 + 4 * pq, N, P, d.O2, d.O2);
    const float *pcol = d.x + (size_t)pc.b * d.x_bstride + pc.p;
    float pq4[2][4][4];       // plain float arrays: arrays of HIP vector structs are not promoted to registers
    unsigned *Bq = reinterpret_cast<unsigned *>(&Bs[0][0][0]);         // [2 stages][4 octets][4 pairs][BN] dwords

#define MI_LOAD_TILE(kt, S)                                                                           \
    do {                                                                                              \
        if (PLAIN) {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                const int c = (kt) * HK + 16 * (i >> 1) + 2 * pg + (i & 1);                           \
                const bool ok = pc.valid && c < d.K;                                                  \
                const float4 t4 = *reinterpret_cast<const float4 *>(ok ? pcol + (size_t)c * P : d.sink + 256);          \
                pq4[S][i][0] = t4.x; pq4[S][i][1] = t4.y; pq4[S][i][2] = t4.z; pq4[S][i][3] = t4.w;                      \
            }                                                                                         \
        } else {                                                                                      \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                           \
                mi_ktab_entry ke[8];                                                                  \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                       \
                    const ktab_i4 e4 = ktab_c[(kt) * HK + 16 * bh + 8 * h + j];                       \
                    ke[j].off = e4.x; ke[j].d1 = e4.y; ke[j].d2 = e4.z; ke[j].ci = e4.w;              \
                }                                                                                     \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                       \
                    bool ok;                                                                          \
                    const float v = gather_b(d, ke[j], xcol, i1b, i2b, lc.valid, ok);                 \
                    breg[0][8 * h + j] = ok ? v : 0.f;                                                \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
    } while (0)
#define MI_LOAD_A(kt)                                                                                 \
    do {                                                                                              \
        if (A_SLOTS >= 1) areg0 = ap0[(size_t)(kt) * a_step];                                            \
        if (A_SLOTS >= 2) areg1 = ap1[(size_t)(kt) * a_step];                                            \
        if (A_SLOTS >= 3) areg2 = ap2[(size_t)(kt) * a_step];                                            \
        if (A_SLOTS >= 4) areg3 = ap3[(size_t)(kt) * a_step];                                            \
    } while (0)

#define MI_STORE_TILE(buf, S)                                                                         \
    do {                                                                                              \
        if (PLAIN) {                                                                                  \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                           \
                const int p = 8 * h + pg;                       /* pair index within the K step */    \
                *reinterpret_cast<uint4 *>(Bq + (((buf) * 4 + (p >> 2)) * 4 + (p & 3)) * BN + 4 * pq) =                   \
                    make_uint4(pack2<HT>(pq4[S][2 * h][0], pq4[S][2 * h + 1][0]), pack2<HT>(pq4[S][2 * h][1], pq4[S][2 * h + 1][1]), \
                               pack2<HT>(pq4[S][2 * h][2], pq4[S][2 * h + 1][2]), pack2<HT>(pq4[S][2 * h][3], pq4[S][2 * h + 1][3])); \
            }                                                                                         \
        } else {                                                                                      \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                             \
                Bs[buf][2 * bh + h][bn] = make_uint4(pack2<HT>(breg[0][8 * h], breg[0][8 * h + 1]), pack2<HT>(breg[0][8 * h + 2], breg[0][8 * h + 3]), \
                                                     pack2<HT>(breg[0][8 * h + 4], breg[0][8 * h + 5]), pack2<HT>(breg[0][8 * h + 6], breg[0][8 * h + 7])); \
        }                                                                                             \
        if (A_SLOTS >= 1 && a_on0) As[buf][ao0][am0] = areg0;                                         \
        if (A_SLOTS >= 2 && a_on1) As[buf][ao1][am1] = areg1;                                         \
        if (A_SLOTS >= 3 && a_on2) As[buf][ao2][am2] = areg2;                                         \
        if (A_SLOTS >= 4 && a_on3) As[buf][ao3][am3] = areg3;                                         \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    // one K step on LDS stage `cur`
#define MI_COMPUTE(cur)                                                                               \
    do {                                                                                              \
        uint4 af[2][TM], bf[2][TN];                                                                   \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                               \
            _Pragma("unroll") for (int a = 0; a < TM; ++a) af[s][a] = As[cur][2 * s + lh][(wm * TM + a) * 32 + li]; \
            _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                          \
                if (PLAIN) {                                                                          \
                    const unsigned *q = Bq + (((cur) * 4 + 2 * s + lh) * 4) * BN + (wn * TN + b) * 32 + li; \
                    bf[s][b] = make_uint4(q[0], q[BN], q[2 * BN], q[3 * BN]);                         \
                } else {                                                                              \
                    bf[s][b] = Bs[cur][2 * s + lh][(wn * TN + b) * 32 + li];                          \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                 \
            _Pragma("unroll") for (int a = 0; a < TM; ++a)                                            \
                _Pragma("unroll") for (int b = 0; b < TN; ++b) acc[a][b] = mfma16<HT>(af[s][a], bf[s][b], acc[a][b]); \
    } while (0)

    // PLAIN: activation tiles are fetched TWO K steps ahead: set 1 holds tile k + 1 while set 0 receives tile k + 2, then
    // set 0 moves into set 1 (16 register moves per step); the gather path and the weights stay one step ahead.
    MI_LOAD_TILE(0, 0);
    MI_LOAD_A(0);
    MI_STORE_TILE(0, 0);
    if (PLAIN && nk > 1) MI_LOAD_TILE(1, 1);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (PLAIN) { if (kt + 2 < nk) MI_LOAD_TILE(kt + 2, 0); }
        else if (kt + 1 < nk) MI_LOAD_TILE(kt + 1, 0);
        if (kt + 1 < nk) MI_LOAD_A(kt + 1);
        MI_COMPUTE(cur);
        if (kt + 1 < nk) { if (PLAIN) MI_STORE_TILE(cur ^ 1, 1); else MI_STORE_TILE(cur ^ 1, 0); }
        if (PLAIN) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) pq4[1][i][e] = pq4[0][i][e];
        }
        __syncthreads();
        cur ^= 1;
    }
#undef MI_COMPUTE
#undef MI_LOAD_TILE
#undef MI_LOAD_A
#undef MI_STORE_TILE
    conv_epilogue<TM, TN, EPI, LFLAGS>(d, acc, m0, n0, wm, wn, N, P, o2v);
}

// ---------------------------------------------------------------------------------------------------------------------
// Plain LINEAR layer whose INPUT exists as a 16-bit operand image (mi_conv_desc.xh: the FFN hidden tensor, written in that
// form by lin1's epilogue): both operand tiles are already in LDS order, so they go global -> LDS with
// global_load_lds_dwordx4 through a 3-stage ring -- TWO K steps in flight behind a counted s_waitcnt vmcnt, one raw
// s_barrier per K step, no staging registers, no conversion (the scheme of conv_gemm_dma_kernel in gemm_conv.hip).
// BM = 64 TM rows; a stage is 4 BM + 512 sixteen-byte words (24 KiB at TM = 4: two workgroups per CU).
typedef __attribute__((address_space(1))) const void hgvoid_t;
typedef __attribute__((address_space(3))) void hlvoid_t;

template <int HT, int TM, int LFLAGS>
__global__ __launch_bounds__(256, 2) void conv_gemm_half_img_kernel(const mi_conv_desc d, const int N, const int MT, const int Gm) {
    constexpr int TN = 2, WN = 2, BM = 2 * TM * 32, NA = BM / 64;
    constexpr int A_W = 4 * BM, SW = A_W + 4 * BN;                  // 16-byte words per stage: A image then B image
    __shared__ __attribute__((aligned(16))) uint4 smem[3 * SW];
    {
        float agpr_anchor = 0.f;
        asm volatile("; accumulators in AGPRs %0" : "+a"(agpr_anchor));
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    if (!tile_of_block(MT, Gm, N, mt, nt)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int P = d.O1 * d.O2;
    const int o2v = d.o2_valid ? d.o2_valid : d.O2;
    const int nk = (d.Kpad + HK - 1) / HK;

    // A: flat [octet][row] image of the K step; wave instruction i = wave * NA + j covers words 64 i .. 64 i + 63
    const uint4 *wh = reinterpret_cast<const uint4 *>(d.wh);
    const uint4 *asrc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int w = 64 * (wave * NA + j) + lane;
        asrc[j] = wh + (size_t)(w / BM) * d.Mpad + m0 + w % BM;
    }
    const size_t a_step = (size_t)4 * d.Mpad;
    // B: this wave moves octet `wave` of the K step, columns [0, 64) and [64, 128) of the tile; columns outside the tensor
    // and octets past K come from the zero page
    const uint4 *xh = reinterpret_cast<const uint4 *>(d.xh);
    const uint4 *zero = reinterpret_cast<const uint4 *>(d.sink + 256);
    const bool c0 = n0 + lane < N, c1 = n0 + 64 + lane < N;
    const uint4 *b0 = xh + (size_t)wave * d.xh_n + n0 + lane;
    const size_t b_step = (size_t)4 * d.xh_n;

#define MI_IMG_TILE(kt, stage)                                                                                      \
    do {                                                                                                            \
        uint4 *sa = smem + (stage) * SW + 64 * (wave * NA), *sb = smem + (stage) * SW + A_W + wave * BN;             \
        _Pragma("unroll") for (int j = 0; j < NA; ++j)                                                               \
            lds_dma16(asrc[j] + (size_t)(kt) * a_step, sa + 64 * j);                                                 \
        const bool kin = ((kt) * 4 + wave) * 8 < d.K;                                                               \
        const uint4 *g = b0 + (size_t)(kt) * b_step;                                                                \
        lds_dma16((kin && c0) ? g : zero, sb);                                                                      \
        lds_dma16((kin && c1) ? g + 64 : zero, sb + 64);                                                            \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    MI_IMG_TILE(0, 0);
    if (nk > 1) MI_IMG_TILE(1, 1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once all but this wave's newest tile (NA + 2 instructions) are done -- for every wave
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // stage (kt + 2) % 3 was last read in the previous iteration, which every wave has finished
        if (kt + 2 < nk) MI_IMG_TILE(kt + 2, stage == 0 ? 2 : stage - 1);
        const uint4 *As = smem + stage * SW, *Bs = As + A_W;
        uint4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[s][a] = As[(2 * s + lh) * BM + (wm * TM + a) * 32 + li];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[s][b] = Bs[(2 * s + lh) * BN + (wn * TN + b) * 32 + li];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = mfma16<HT>(af[s][a], bf[s][b], acc[a][b]);
        stage = stage == 2 ? 0 : stage + 1;
    }
#undef MI_IMG_TILE
    conv_epilogue<TM, TN, MI_EPI_LINEAR, LFLAGS>(d, acc, m0, n0, wm, wn, N, P, o2v);
}


// ---------------------------------------------------------------------------------------------------------------------
// The same layer on a 256 x 256 tile with EIGHT waves (2 x 4, 128 x 64 outputs each) and a FOUR-stage ring: at 256 x 128 a K step
// moves 24 KiB global -> LDS for 2.1 MFLOP, which at the matrix pipe's rate is ~48 bytes per cycle and CU -- more than the
// L2 -> LDS path delivers (~33, MI355X_MICROARCH.md "Indexed rows") and the SQ counters show the waves parked on vmcnt /
// barrier 55 % of their cycles.  256 x 256 needs 32 KiB per K step for twice the work (32 B per cycle and CU), keeps three
// K steps in flight behind a counted vmcnt, and issues its DMA with a wave-uniform base (no VALU slot on addresses).  One
// workgroup of 512 threads per CU (128 KiB of LDS), two waves per SIMD.
template <int HT, int LFLAGS>
__global__ __launch_bounds__(512, 1) void conv_gemm_half_img256_kernel(const mi_conv_desc d, const int N, const int MT, const int NT) {
    constexpr int TM = 4, TN = 2, WN = 4, BM = 256, BN2 = 256, NST = 4;
    constexpr int A_W = 4 * BM, SW = A_W + 4 * BN2;               // 16-byte words per stage: A image then B image (32 KiB)
    __shared__ __attribute__((aligned(16))) uint4 smem[NST * SW];
    {
        float agpr_anchor = 0.f;
        asm volatile("; accumulators in AGPRs %0" : "+a"(agpr_anchor));
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware order: workgroups id, id + 8, ... share an L2; XCD x walks the N tiles nt = x (mod 8), the M tiles of one N
    // tile back to back (they share the activation tile; the weights stay L2 resident)
    const int xj = blockIdx.x >> 3;
    const int nt = (xj / MT) * 8 + (blockIdx.x & 7), mt = xj % MT;
    if (nt >= NT) return;                                          // grid padding (whole workgroup, before any barrier)
    const int m0 = mt * BM, n0 = nt * BN2;
    const int P = d.O1 * d.O2;
    const int o2v = d.o2_valid ? d.o2_valid : d.O2;
    const int nk = (d.Kpad + HK - 1) / HK;

    // A: K step image = 4 octets x 256 rows = 16 wave instructions of 64 words; wave w issues 2 w and 2 w + 1 (octet w / 2,
    // rows 128 (w & 1) + 64 j); B: the same over 256 columns.  Bases are wave-uniform, the lane offset is 16 lane.
    const uint4 *wh = reinterpret_cast<const uint4 *>(d.wh);
    const uint4 *xh = reinterpret_cast<const uint4 *>(d.xh);
    const uint4 *zero = reinterpret_cast<const uint4 *>(d.sink + 256);
    const int oct = wave >> 1, half = wave & 1;
    const uint4 *abase = wh + (size_t)oct * d.Mpad + m0 + 128 * half;
    const uint4 *bbase = xh + (size_t)oct * d.xh_n + n0 + 128 * half;
    const size_t a_step = (size_t)4 * d.Mpad, b_step = (size_t)4 * d.xh_n;
    const bool full_n = n0 + BN2 <= N;                              // wave-uniform: the ragged last column tile takes per-lane sources
    const unsigned loff = 16u * lane;
#define MI_IMG2_TILE(kt, stage)                                                                                      \
    do {                                                                                                             \
        uint4 *sa = smem + (stage) * SW + oct * BM + 128 * half, *sb = smem + (stage) * SW + A_W + oct * BN2 + 128 * half; \
        const uint4 *ga = abase + (size_t)(kt) * a_step, *gb = bbase + (size_t)(kt) * b_step;                       \
        lds_dma16_s(ga, loff, sa);                                                                                   \
        lds_dma16_s(ga + 64, loff, sa + 64);                                                                         \
        const bool kin = ((kt) * 4 + oct) * 8 < d.K;                                                                 \
        if (full_n && kin) {                                                                                         \
            lds_dma16_s(gb, loff, sb);                                                                               \
            lds_dma16_s(gb + 64, loff, sb + 64);                                                                     \
        } else {                                                                                                     \
            const int c_ = n0 + 128 * half + lane;                                                                   \
            lds_dma16((kin && c_ < N) ? gb + lane : zero, sb);                                                       \
            lds_dma16((kin && c_ + 64 < N) ? gb + 64 + lane : zero, sb + 64);                                        \
        }                                                                                                            \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    MI_IMG2_TILE(0, 0);
    if (nk > 1) MI_IMG2_TILE(1, 1);
    if (nk > 2) MI_IMG2_TILE(2, 2);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once all but this wave's two newest tiles (4 instructions each) are done -- for every wave
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // stage (kt + 3) % 4 was last read in the previous iteration, which every wave has finished
        if (kt + 3 < nk) MI_IMG2_TILE(kt + 3, (stage + 3) & 3);
        const uint4 *As = smem + stage * SW, *Bs = As + A_W;
        uint4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[s][a] = As[(2 * s + lh) * BM + (wm * TM + a) * 32 + li];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[s][b] = Bs[(2 * s + lh) * BN2 + (wn * TN + b) * 32 + li];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = mfma16<HT>(af[s][a], bf[s][b], acc[a][b]);
        stage = (stage + 1) & 3;
    }
#undef MI_IMG2_TILE
    conv_epilogue<TM, TN, MI_EPI_LINEAR, LFLAGS>(d, acc, m0, n0, wm, wn, N, P, o2v);
}

template <int HT, int LFLAGS>
static int launch_half_img256(const mi_conv_desc &d, hipStream_t st) {
    const int64_t N64 = (int64_t)d.B * d.O1 * d.O2;
    MI_REQUIRE(N64 < (1ll << 31) - 256 && N64 <= d.xh_n, "conv: %lld output positions, operand image has %lld columns", (long long)N64,
               (long long)d.xh_n);
    MI_REQUIRE(d.Mpad % 256 == 0 && d.K % 8 == 0 && ((uintptr_t)d.xh & 15) == 0, "conv: operand-image layer (256-row tile) needs Mpad %% 256 == 0, K %% 8 == 0");
    const int N = (int)N64, MT = d.Mpad / 256, NT = ceil_div(N, 256);
    hipLaunchKernelGGL((conv_gemm_half_img256_kernel<HT, LFLAGS>), dim3((unsigned)(8 * MT * ((NT + 7) / 8))), dim3(512), 0, st, d, N, MT, NT);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int HT, int TM, int LFLAGS>
static int launch_half_img(const mi_conv_desc &d, hipStream_t st) {
    constexpr int BM = 2 * TM * 32;
    const int64_t N64 = (int64_t)d.B * d.O1 * d.O2;
    MI_REQUIRE(N64 < (1ll << 31) - 256 && N64 <= d.xh_n, "conv: %lld output positions, operand image has %lld columns", (long long)N64,
               (long long)d.xh_n);
    MI_REQUIRE(d.Mpad % BM == 0 && d.K % 8 == 0 && ((uintptr_t)d.xh & 15) == 0, "conv: operand-image layer needs Mpad %% %d == 0, K %% 8 == 0", BM);
    const int N = (int)N64, MT = d.Mpad / BM, NT = ceil_div(N, BN);
    hipLaunchKernelGGL((conv_gemm_half_img_kernel<HT, TM, LFLAGS>), dim3(grouped_grid(MT, NT, 1)), dim3(256), 0, st, d, N, MT, 1);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int HT, int WM, int WN, int TM, int TN, int EPI, int LFLAGS, bool PLAIN>
static int launch_cfg_half(const mi_conv_desc &d, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    const int64_t N64 = (int64_t)d.B * d.O1 * d.O2;
    MI_REQUIRE(N64 < (1ll << 31) - 256, "conv: too many output positions (%lld)", (long long)N64);
    MI_REQUIRE(d.Mpad % BM == 0, "conv: Mpad %d not a multiple of the %d-row tile", d.Mpad, BM);
    const int N = (int)N64, MT = d.Mpad / BM, NT = ceil_div(N, BN);
    const int Gm = 1;
    const unsigned grid = grouped_grid(MT, NT, Gm);
    hipLaunchKernelGGL((conv_gemm_half_kernel<HT, WM, WN, TM, TN, EPI, LFLAGS, PLAIN>), dim3(grid), dim3(256), 0, st, d, N, MT, Gm);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int HT, int EPI, int LFLAGS, bool PLAIN>
static int launch_tile_half(const mi_conv_desc &d, int tile, hipStream_t st) {
    if constexpr (PLAIN && EPI == MI_EPI_LINEAR) {
        // 256 x 128 for the transformer's linear layers: the 128 x 128 tile is bound by cache bandwidth (43 FLOP per byte of
        // float32 activations + 16-bit weights); twice the rows per activation tile brings that to 64
        if (tile == 256) return launch_cfg_half<HT, 2, 2, 4, 2, EPI, LFLAGS, PLAIN>(d, st);
    }
    switch (tile) {
        case 128: return launch_cfg_half<HT, 2, 2, 2, 2, EPI, LFLAGS, PLAIN>(d, st);
        case 96: return launch_cfg_half<HT, 1, 4, 3, 1, EPI, LFLAGS, PLAIN>(d, st);
        case 64: return launch_cfg_half<HT, 1, 4, 2, 1, EPI, LFLAGS, PLAIN>(d, st);
        case 32: return launch_cfg_half<HT, 1, 4, 1, 1, EPI, LFLAGS, PLAIN>(d, st);
    }
    return set_error(MI_EINVAL, "conv half: unsupported tile_m %d", tile);
}

template <int HT>
static int launch_conv_half_t(const mi_conv_desc &d, int tile, bool plain, hipStream_t st) {
#define MI_DISPATCH(E)                                              \
    case E: return plain ? launch_tile_half<HT, E, 0, true>(d, tile, st) : launch_tile_half<HT, E, 0, false>(d, tile, st)
#define MI_LINEAR(F)                                                \
    case F: return plain ? launch_tile_half<HT, MI_EPI_LINEAR, F, true>(d, tile, st) : launch_tile_half<HT, MI_EPI_LINEAR, F, false>(d, tile, st)
    if (d.xh) {         // input given as a 16-bit operand image: lin2 / out_proj, and (256-row tile) the in-projections and lin1
        const int f = d.flags & (MI_FLAG_GELU | MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_LN | MI_FLAG_IMG | MI_FLAG_HEADS | MI_FLAG_STATS);
        MI_REQUIRE(d.epi == MI_EPI_LINEAR && plain, "conv: an operand-image input needs a plain LINEAR layer");
        // residual epilogues (out_proj, lin2: 340 MB of float32 residual read + output write per launch) keep the 256 x 128 tile
        // at two workgroups per CU, whose epilogues overlap each other's main loops: measured 9.63 ms against 9.79 for the
        // linear class with the 256 x 256 tile here (MI_IMG256=1: A/B switch)
        static const bool img256 = [] { const char *e = getenv("MI_IMG256"); return e && atoi(e) != 0; }();
        if (f == (MI_FLAG_SCALE | MI_FLAG_RES)) {
            if (img256 && d.Mpad % 256 == 0) return launch_half_img256<HT, MI_FLAG_SCALE | MI_FLAG_RES>(d, st);
            return d.Mpad % 256 == 0 ? launch_half_img<HT, 4, MI_FLAG_SCALE | MI_FLAG_RES>(d, st) : launch_half_img<HT, 2, MI_FLAG_SCALE | MI_FLAG_RES>(d, st);
        }
        if (f == (MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_STATS))
            return d.Mpad % 256 == 0 ? launch_half_img<HT, 4, MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_STATS>(d, st)
                                     : launch_half_img<HT, 2, MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_STATS>(d, st);
        if (f == (MI_FLAG_LN | MI_FLAG_HEADS)) return launch_half_img256<HT, MI_FLAG_LN | MI_FLAG_HEADS>(d, st);
        if (f == (MI_FLAG_LN | MI_FLAG_GELU | MI_FLAG_IMG)) return launch_half_img256<HT, MI_FLAG_LN | MI_FLAG_GELU | MI_FLAG_IMG>(d, st);
        return set_error(MI_EINVAL, "conv: an operand-image input is not instantiated for LINEAR flags %d", d.flags);
    }
    if (d.epi == MI_EPI_LINEAR) {
        switch (d.flags & (MI_FLAG_GELU | MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_LN | MI_FLAG_IMG | MI_FLAG_HEADS | MI_FLAG_STATS)) {
            MI_LINEAR(MI_FLAG_LN | MI_FLAG_GELU | MI_FLAG_IMG);
            MI_LINEAR(MI_FLAG_LN | MI_FLAG_HEADS);
            MI_LINEAR(0);
            MI_LINEAR(MI_FLAG_GELU);
            MI_LINEAR(MI_FLAG_RES);
            MI_LINEAR(MI_FLAG_RES | MI_FLAG_IMG);
            MI_LINEAR(MI_FLAG_SCALE | MI_FLAG_RES);
            MI_LINEAR(MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_STATS);
            MI_LINEAR(MI_FLAG_LN);
            MI_LINEAR(MI_FLAG_LN | MI_FLAG_GELU);
        }
        return set_error(MI_EINVAL, "conv: unsupported LINEAR flag combination %d", d.flags);
    }
#undef MI_LINEAR
    switch (d.epi) {
        MI_DISPATCH(MI_EPI_GLU);
        MI_DISPATCH(MI_EPI_BIAS_STATS);
        MI_DISPATCH(MI_EPI_STATS_ONLY);
        MI_DISPATCH(MI_EPI_GN_GLU);
        MI_DISPATCH(MI_EPI_CONVTR);
    }
#undef MI_DISPATCH
    return set_error(MI_EINVAL, "conv: unsupported epilogue %d", d.epi);
}

// d has been validated by launch_conv (gemm_conv.hip), which also decided `plain`
int launch_conv_half(const mi_conv_desc &d, int tile, bool plain, hipStream_t st) {
    MI_REQUIRE(d.wh && ((uintptr_t)d.wh & 15) == 0, "conv half: weight image missing or misaligned");
    MI_REQUIRE(plain || d.ktab_len >= (d.Kpad + HK - 1) / HK * HK, "conv half: gather table has %d entries, the K step of %d needs %d",
               d.ktab_len, HK, (d.Kpad + HK - 1) / HK * HK);
    static const bool wide = [] { const char *e = getenv("MI_HALF_TILE256"); return !e || atoi(e) != 0; }();
    if (wide && plain && tile == 128 && d.epi == MI_EPI_LINEAR && d.Mpad % 256 == 0) tile = 256;
    if (d.half == MI_DTYPE_BF16) return launch_conv_half_t<MI_DTYPE_BF16>(d, tile, plain, st);
    if (d.half == MI_DTYPE_F16) return launch_conv_half_t<MI_DTYPE_F16>(d, tile, plain, st);
    return set_error(MI_EINVAL, "conv half: operand type %d", d.half);
}

}  // namespace mi
