// Half modes: stride-1 k x k convolutions whose INPUT exists as a 16-bit operand image  X[Cin / 8][positions][8]:
//   * the decoders' 3 x 3 / k = 3 "rewrite" convs (reference demucs/hdemucs.py:304-314): their input is written ONLY as that
//     image by the producing epilogue (the previous layer's transposed conv + GELU + skip, or the channel down-sampler);
//   * the decoders' transposed convs (hdemucs.py:287,326-334) as s-phase GEMMs with two taps (input q and q - 1): their
//     input, the DConv branch's float32 output, is converted to an image by one streaming pass (f32_to_image_kernel);
//   * the encoders' strided convs of levels 1-3 (k = 8, s = 4, pad 2; hdemucs.py:110,132-136): on the PHASE-SPLIT image the
//     previous level's 1x1 + GLU epilogue writes (gemm_conv.h MI_FLAG_IMG4) they are stride-1 two-tap convs over 4 Cin channels.
//
// K is enumerated TAP-MINOR PER CHANNEL OCTET:  k = ((ci / 8) * ntaps + tap) * 8 + ci % 8.  One LDS row of a K step's B tile
// (eight consecutive k = eight channels of ONE tap, for 128 consecutive output positions) is then a contiguous run of the
// image shifted by the tap's offset d1 * D2 + d2 -- so the implicit-GEMM gather becomes two LDS-DMA wave instructions per
// row with per-lane sources (the zero page where the tap falls outside the frame), exactly like the plain image layers of
// gemm_half.hip: no float32 load, no conversion, no table walk in the loop.  The table-driven loader of gemm_half.hip
// spends ~200 VALU slots per K step and wave on 16 dword gathers, their bounds tests and the rounding (SQ counters: 45-50 %
// of the wave cycles parked on vmcnt, 20 % VALU) for 16 matrix instructions.
// Weights: the same packed rows (GLU interleave included) with k permuted to that order (mi_conv_pack_tap).
// Three-stage ring, counted vmcnt, one barrier per K step; A rows are padded to 128 per octet in LDS so that every wave
// issues the same number of DMA instructions whatever the tile height.
#include "gemm_tile.h"

namespace mi {

namespace {
typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tbf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 tf16x2 __attribute__((ext_vector_type(2)));

template <int HT>
__device__ __forceinline__ f32x16 tmfma(const uint4 a, const uint4 b, const f32x16 c) {
    if (HT == MI_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tbf16x8, a), __builtin_bit_cast(tbf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tf16x8, a), __builtin_bit_cast(tf16x8, b), c, 0, 0, 0);
}
}  // namespace

// Wt[Kpad][Mpad] fp32 (k = ci * ntaps + tap) -> Wtap[pairs rounded up to 4][Mpad][8], pair = (ci / 8) * ntaps + tap
template <int HT>
__global__ void pack_tap_kernel(const float *__restrict__ wt, int Mpad, int Cin, int ntaps, int pairs_pad, unsigned short *__restrict__ out) {
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t total = (size_t)pairs_pad * Mpad * 8;
    if (idx >= total) return;
    const int j = (int)(idx & 7), m = (int)((idx >> 3) % Mpad), p = (int)((idx >> 3) / Mpad);
    const int co = p / ntaps, tap = p - co * ntaps, ci = 8 * co + j;
    const float x = ci < Cin ? wt[(size_t)(ci * ntaps + tap) * Mpad + m] : 0.f;
    unsigned bits;
    if (HT == MI_DTYPE_BF16) { const tbf16x2 h = {(__bf16)x, (__bf16)0.f}; bits = __builtin_bit_cast(unsigned, h); }
    else { const tf16x2 h = {(_Float16)x, (_Float16)0.f}; bits = __builtin_bit_cast(unsigned, h); }
    out[idx] = (unsigned short)(bits & 0xffffu);
}

int conv_tap_pairs_pad(int Cin, int ntaps) { return (Cin / 8 * ntaps + 3) / 4 * 4; }

int launch_pack_tap(const float *wt, int Mpad, int Cin, int ntaps, int dtype, void *out, hipStream_t st) {
    MI_REQUIRE((dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16) && Cin % 8 == 0 && ntaps >= 1, "pack_tap: dtype %d, Cin %d, taps %d", dtype, Cin, ntaps);
    const int pp = conv_tap_pairs_pad(Cin, ntaps);
    const size_t n = (size_t)pp * Mpad * 8;
    if (dtype == MI_DTYPE_BF16)
        hipLaunchKernelGGL(pack_tap_kernel<MI_DTYPE_BF16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wt, Mpad, Cin, ntaps, pp, (unsigned short *)out);
    else
        hipLaunchKernelGGL(pack_tap_kernel<MI_DTYPE_F16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wt, Mpad, Cin, ntaps, pp, (unsigned short *)out);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int HT, int WM, int WN, int TM, int TN, int EPI, int LFLAGS>
// Three workgroups per CU (round 4: the bound was 2 and hipcc took 192-196 registers for the 128-row tiles; at 168 nothing goes to
// scratch): the 128-row GLU class 4.33 -> 3.80 ms and the transposed convs 3.17 -> 2.77 ms per three steps, bf16 step -0.3 ms.
__global__ __launch_bounds__(256, 3) void conv_gemm_half_tap_kernel(const mi_conv_desc d, const int N, const int MT, const int Gm) {
    constexpr int BM = WM * TM * 32, AP = 128;                  // A rows per octet in LDS (padded: two DMA instructions per octet)
    static_assert(WN * TN * 32 == BN && WM * WN == 4 && BM <= AP, "4 waves, 128 columns, at most 128 rows");
    constexpr int SW = 4 * AP + 4 * BN;                          // 16-byte words per stage (16 KiB)
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
    const int ntaps = d.ntaps, npairs = d.K / 8;                 // K = Cin * ntaps, Cin % 8 == 0
    const int nk = (npairs + 3) / 4;

    const uint4 *wimg = reinterpret_cast<const uint4 *>(d.wtap);
    const uint4 *ximg = reinterpret_cast<const uint4 *>(d.xh);
    const uint4 *zero = reinterpret_cast<const uint4 *>(d.sink + 256);
    // this wave moves octet `wave` of every K step: A rows m0 + 64 j + lane, B columns n0 + 64 j + lane (j = 0, 1)
    const bool arow0 = m0 + lane < d.Mpad && lane < BM, arow1 = m0 + 64 + lane < d.Mpad && 64 + lane < BM;
    // output position of the two columns: (b, o1, o2); the input position of tap (d1, d2) is (o1 + d1, o2 + d2), stride 1
    int cb[2], co1[2], co2[2];
    bool cok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const ColInfo c = decompose(n0 + 64 * j + lane, N, P, d.O2, o2v);
        cb[j] = c.b; co1[j] = c.o1; co2[j] = c.o2; cok[j] = c.valid;
    }
    const int x_ld = d.x_ld ? d.x_ld : d.D2;                     // row pitch of the input positions (= O2 of this stride-1 conv)
    const int64_t npos = d.xh_n;                                 // positions per channel octet of the image
    // (channel octet, tap) of this wave's octet, advanced by four pairs per K step (scalar arithmetic)
    const int K2 = d.tap_k2, K1 = ntaps / K2, dil1 = d.tap_dil1 ? d.tap_dil1 : 1, dil2 = d.tap_dil2 ? d.tap_dil2 : 1;
    int p_idx = wave, p_oct = wave / ntaps, p_t1 = (wave - p_oct * ntaps) / K2, p_t2 = (wave - p_oct * ntaps) - p_t1 * K2;

#define MI_TAP_TILE(stage)                                                                                            \
    do {                                                                                                              \
        uint4 *sa = smem + (stage) * SW + wave * AP, *sb = smem + (stage) * SW + 4 * AP + wave * BN;                  \
        const bool pin = p_idx < npairs;                                                                              \
        const uint4 *ga = wimg + (size_t)p_idx * d.Mpad + m0 + lane;                                                  \
        lds_dma16((pin && arow0) ? ga : zero, sa);                                                                    \
        lds_dma16((pin && arow1) ? ga + 64 : zero, sa + 64);                                                          \
        const int d1 = p_t1 * dil1 - d.tap_pad1, d2 = p_t2 * dil2 - d.tap_pad2;                                       \
        const uint4 *gx = ximg + (size_t)p_oct * npos;                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                               \
            const int i1 = co1[j] + d1, i2 = co2[j] + d2;                                                             \
            const bool ok = pin && cok[j] && (unsigned)i1 < (unsigned)d.D1 && (unsigned)i2 < (unsigned)d.D2;          \
            const uint4 *src = gx + ((size_t)cb[j] * d.D1 + i1) * x_ld + i2;                                          \
            lds_dma16(ok ? src : zero, sb + 64 * j);                                                                  \
        }                                                                                                             \
        p_idx += 4; p_t2 += 4;                                                                                        \
        while (p_t2 >= K2) { p_t2 -= K2; ++p_t1; }                                                                    \
        while (p_t1 >= K1) { p_t1 -= K1; ++p_oct; }                                                                   \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    MI_TAP_TILE(0);
    if (nk > 1) MI_TAP_TILE(1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once all but this wave's newest tile (4 instructions) are done -- for every wave after the barrier
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // stage (kt + 2) % 3 was last read in the previous iteration, which every wave has finished
        if (kt + 2 < nk) MI_TAP_TILE(stage == 0 ? 2 : stage - 1);
        const uint4 *As = smem + stage * SW, *Bs = As + 4 * AP;
        uint4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[s][a] = As[(2 * s + lh) * AP + (wm * TM + a) * 32 + li];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[s][b] = Bs[(2 * s + lh) * BN + (wn * TN + b) * 32 + li];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = tmfma<HT>(af[s][a], bf[s][b], acc[a][b]);
        stage = stage == 2 ? 0 : stage + 1;
    }
#undef MI_TAP_TILE
    conv_epilogue<TM, TN, EPI, LFLAGS>(d, acc, m0, n0, wm, wn, N, P, o2v);
}

// x[b][C][P] float32 (channel stride P, batch stride C P) -> image [C / 8][B P][8] in one streaming pass (coalesced along positions)
template <int HT>
__global__ __launch_bounds__(256) void f32_to_image_kernel(const float *__restrict__ x, int C, int64_t P, int64_t total, uint4 *__restrict__ img) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= total) return;
    const int oct = blockIdx.y;
    const int64_t b = n / P, p = n - b * P;
    const float *src = x + ((size_t)b * C + 8 * oct) * P + p;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)j * P];
    img[(size_t)oct * total + n] = make_uint4(pack_half2(HT, v[0], v[1]), pack_half2(HT, v[2], v[3]), pack_half2(HT, v[4], v[5]), pack_half2(HT, v[6], v[7]));
}

int launch_f32_to_image(const float *x, int B, int C, int64_t P, int dtype, void *img, hipStream_t st) {
    MI_REQUIRE(C % 8 == 0 && (dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16) && ((uintptr_t)img & 15) == 0, "to_image: C %d, dtype %d", C, dtype);
    const int64_t total = (int64_t)B * P;
    const dim3 grid((unsigned)((total + 255) / 256), C / 8);
    if (dtype == MI_DTYPE_BF16) hipLaunchKernelGGL(f32_to_image_kernel<MI_DTYPE_BF16>, grid, dim3(256), 0, st, x, C, P, total, (uint4 *)img);
    else hipLaunchKernelGGL(f32_to_image_kernel<MI_DTYPE_F16>, grid, dim3(256), 0, st, x, C, P, total, (uint4 *)img);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int HT, int WM, int WN, int TM, int TN, int EPI, int LFLAGS>
static int launch_tap_cfg(const mi_conv_desc &d, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    const int64_t N64 = (int64_t)d.B * d.O1 * d.O2;
    MI_REQUIRE(N64 < (1ll << 31) - 256, "conv tap: too many output positions (%lld)", (long long)N64);
    MI_REQUIRE(d.Mpad % BM == 0, "conv tap: Mpad %d not a multiple of the %d-row tile", d.Mpad, BM);
    const int N = (int)N64, MT = d.Mpad / BM, NT = ceil_div(N, BN);
    hipLaunchKernelGGL((conv_gemm_half_tap_kernel<HT, WM, WN, TM, TN, EPI, LFLAGS>), dim3(grouped_grid(MT, NT, 1)), dim3(256), 0, st, d, N, MT, 1);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// d validated by launch_conv (gemm_conv.hip): half mode, stride-1 conv (the strided encoder convs arrive as stride-1 two-tap convs on
// a phase-split image, gemm_conv.h MI_FLAG_IMG4), input image + tap-ordered weights; epilogues GLU, transposed-conv scatter, bias + GELU
int launch_conv_tap(const mi_conv_desc &d, int tile, hipStream_t st) {
    const int lflags = d.flags & (MI_FLAG_GELU | MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_LN | MI_FLAG_IMG | MI_FLAG_HEADS | MI_FLAG_STATS);
    const bool lin_gelu = d.epi == MI_EPI_LINEAR && lflags == MI_FLAG_GELU;
    const bool lin_bias = d.epi == MI_EPI_LINEAR && lflags == 0;        // hdemucs layers 4 / 5: k = 3 convs followed by a GroupNorm pass
    MI_REQUIRE((d.epi == MI_EPI_GLU || d.epi == MI_EPI_CONVTR || lin_gelu || lin_bias) && d.S1 == 1 && d.S2 == 1,
               "conv tap: instantiated for stride-1 GLU convs, transposed convs and bias (+ GELU) convs");
    MI_REQUIRE(d.wtap && d.xh && d.ntaps >= 1 && d.tap_k2 >= 1 && d.K % (8 * d.ntaps) == 0 && (((uintptr_t)d.wtap | (uintptr_t)d.xh) & 15) == 0,
               "conv tap: needs the tap-ordered weight image, the input image and K = Cin * ntaps with Cin %% 8 == 0");
    MI_REQUIRE(d.xh_n >= (int64_t)d.B * d.D1 * (d.x_ld ? d.x_ld : d.D2), "conv tap: input image has %lld positions", (long long)d.xh_n);
#define MI_TAP_E(W1, W2, T1, T2, E, F) (d.half == MI_DTYPE_BF16 ? launch_tap_cfg<MI_DTYPE_BF16, W1, W2, T1, T2, E, F>(d, st) : launch_tap_cfg<MI_DTYPE_F16, W1, W2, T1, T2, E, F>(d, st))
#define MI_TAP(W1, W2, T1, T2) (d.epi == MI_EPI_GLU ? MI_TAP_E(W1, W2, T1, T2, MI_EPI_GLU, 0) : d.epi == MI_EPI_CONVTR ? MI_TAP_E(W1, W2, T1, T2, MI_EPI_CONVTR, 0) : lin_bias ? MI_TAP_E(W1, W2, T1, T2, MI_EPI_LINEAR, 0) : MI_TAP_E(W1, W2, T1, T2, MI_EPI_LINEAR, MI_FLAG_GELU))
    switch (tile) {
        case 128: return MI_TAP(2, 2, 2, 2);
        case 96: return MI_TAP(1, 4, 3, 1);
        case 64: return MI_TAP(1, 4, 2, 1);
        case 32: return MI_TAP(1, 4, 1, 1);
    }
#undef MI_TAP
#undef MI_TAP_E
    return set_error(MI_EINVAL, "conv tap: unsupported tile_m %d", tile);
}

}  // namespace mi
