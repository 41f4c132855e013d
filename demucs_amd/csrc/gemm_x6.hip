// Implicit-GEMM convolution with fp32 operands carried as THREE bf16 terms each (x = hi + mid + lo, exact: 3 x 8
// mantissa bits = the 24 of fp32) and SIX bf16 MFMA products accumulated in fp32:
//
//     a*b  ~=  ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh          (dropped terms <= 2^-24 |a b|)
//
// Every product of two bf16 values is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the
// result carries the same ~2^-24 relative rounding per term as the native fp32 MFMA path of gemm_conv.hip
// (measured against the fp64 oracle: tests/test_gpu_kernels.py::test_conv_x6_*), while the matrix pipe runs
// bf16 at 16x the fp32 rate: 6 products = 2.67x fewer matrix-pipe cycles per fp32-equivalent FLOP.
//
// Tile image in LDS, per operand and K step of 16: [part 3][k-half 2][row or column][8 bf16] -- a 32x32x16
// fragment (lane l: row l&31, k = 8*(l>>5) .. +7) is ONE conflict-free ds_read_b128 per part.
//   A (weights): split once at load time into exactly this image per (M tile, K step) (pack_split_kernel),
//                so the loader is pure LDS-DMA (global_load_lds_dwordx4, no registers, no ds_write).
//   B (activations): thread (column n = tid & 127, k-half = tid >> 7) loads its 8 k values (coalesced dwords,
//                through the same gather table as the fp32 kernel), splits them in registers (v_cvt_pk_bf16_f32,
//                round-to-nearest residuals) and writes three 16-byte words.
// Epilogues are the shared ones of gemm_tile.h (the accumulator layout of all 32x32 MFMAs is the same).
#include "gemm_tile.h"

namespace mi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// (x0, x1) -> packed bf16 pairs of the three terms; x == hi + mid + lo exactly (each residual is representable)
__device__ __forceinline__ void split3(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    const bf16x2 hh = {(__bf16)x0, (__bf16)x1};
    const float r0 = x0 - (float)hh[0], r1 = x1 - (float)hh[1];
    const bf16x2 mm = {(__bf16)r0, (__bf16)r1};
    const float q0 = r0 - (float)mm[0], q1 = r1 - (float)mm[1];
    const bf16x2 ll = {(__bf16)q0, (__bf16)q1};
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, mm);
    l = __builtin_bit_cast(unsigned, ll);
}

// Wt[Kpad][Mpad] fp32 -> Wx[mt][kt][part][k-half][BM rows][8] bf16
__global__ void pack_split_kernel(const float *__restrict__ wt, int Kpad, int Mpad, int BM, __bf16 *__restrict__ wx) {
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= (size_t)Kpad * Mpad) return;
    const int k = (int)(idx / Mpad), m = (int)(idx % Mpad);
    const int mt = m / BM, mi = m % BM, kt = k / BK, h = (k % BK) / 8, j = k % 8, nk = Kpad / BK;
    const float x = wt[idx];
    const __bf16 a = (__bf16)x;
    const float r = x - (float)a;
    const __bf16 b = (__bf16)r;
    const __bf16 c = (__bf16)(r - (float)b);
    __bf16 *img = wx + ((size_t)mt * nk + kt) * ((size_t)BM * 48);
    img[((0 * 2 + h) * BM + mi) * 8 + j] = a;
    img[((1 * 2 + h) * BM + mi) * 8 + j] = b;
    img[((2 * 2 + h) * BM + mi) * 8 + j] = c;
}

int launch_pack_split(const float *wt, int Kpad, int Mpad, int tile_m, void *wx, hipStream_t st) {
    MI_REQUIRE(Kpad % BK == 0 && Mpad % tile_m == 0, "pack_split: Kpad %d / Mpad %d do not fit tile %d", Kpad, Mpad, tile_m);
    const size_t n = (size_t)Kpad * Mpad;
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wt, Kpad, Mpad, tile_m, (__bf16 *)wx);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ACCUMULATORS IN AGPRs.  On this platform a stream of v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 whose destination is in
// ARCHITECTURAL VGPRs corrupts kernels of OTHER processes that run on the same CUs (tools/micro/mfma_neighbour.hip:
// a bare MFMA loop with no memory traffic is enough; fp32 MFMAs and bf16 MFMAs with AGPR accumulators -- what the
// vendor GEMMs use -- are clean).  hipcc picks the VGPR form whenever the registers fit; one inline-asm AGPR operand in
// the kernel makes it allocate AGPRs and place every MFMA accumulator there.
#ifndef MI_X6_ABL
#define MI_X6_ABL 0          /* victim-side bisect builds: 1 = one MFMA per K step instead of 24, 4 = no epilogue, 8 = zero operands,
                                16 = 32 idle cycles after every MFMA, 32 = 64 idle cycles after every 4 MFMAs */
#endif
template <int WM, int WN, int TM, int TN, int EPI, int LFLAGS, bool PLAIN>
__global__ __launch_bounds__(256, 2) void conv_gemm_x6_kernel(const mi_conv_desc d, const int N, const int MT, const int Gm) {
    constexpr int BM = WM * TM * 32;
    static_assert(WN * TN * 32 == BN, "block N tile is 128");
    static_assert(WM * WN == 4, "4 waves");
    constexpr int A_BYTES = BM * 96, B_BYTES = BN * 96, STAGE = A_BYTES + B_BYTES;
    constexpr int A_CHUNKS = A_BYTES / 1024;                 // 1 KiB per wave-wide DMA instruction
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    // plain layers: the fp32 activation tile [16][128] lands here by LDS-DMA two K steps ahead and is split from LDS
    // (a VMEM wave-instruction costs ~12-16 cycles whatever its width: 2 x 1 KiB DMA per wave replace 8 dword loads)
    __shared__ __attribute__((aligned(16))) float braw[PLAIN ? 2 * BK * BN : 4];

    {   // keeps the MFMA accumulators in AGPRs (see the note above the kernel); the Makefile builds this file with
        // -amdgpu-mfma-vgpr-form=0
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
    const int nk = d.Kpad / BK;

    // ---- A: LDS-DMA of the pre-split image; wave w moves chunks w, w + 4, ... -------------------------
    const unsigned char *aimg = reinterpret_cast<const unsigned char *>(d.wx) + (size_t)mt * nk * A_BYTES + lane * 16;
#define MI_A_DMA(kt, stage)                                                                                         \
    do {                                                                                                            \
        _Pragma("unroll") for (int c = 0; c < (A_CHUNKS + 3) / 4; ++c) {                                             \
            const int chunk = c * 4 + wave;                                                                         \
            if (chunk < A_CHUNKS)                                                                                   \
                __builtin_amdgcn_global_load_lds((gvoid_t *)(aimg + (size_t)(kt) * A_BYTES + chunk * 1024),         \
                                                 (lvoid_t *)(smem + (stage) * STAGE + chunk * 1024), 16, 0, 0);     \
        }                                                                                                           \
    } while (0)

    // ---- B: this thread owns column bn and the 8 k values of k-half bh ----------------------------------
    const int bn = tid & 127;
    const int bh = __builtin_amdgcn_readfirstlane(tid >> 7);
    const ColInfo lc = decompose(n0 + bn, N, P, d.O2, PLAIN ? d.O2 : o2v);
    const int i1b = lc.o1 * d.S1, i2b = lc.o2 * d.S2;
    const float *xcol = d.x + (size_t)lc.b * d.x_bstride + (PLAIN ? (size_t)lc.p : (size_t)i1b * (d.x_ld ? d.x_ld : d.D2) + i2b);
    const float *bp = lc.valid ? xcol + (size_t)(8 * bh) * P : d.sink + 256;     // plain: channel stride = P
    const size_t b_row = lc.valid ? (size_t)P : 0, b_step = lc.valid ? (size_t)BK * P : 0;
    float breg[8];
    // plain DMA loader: this lane's 16 bytes of rows 4*wave + 2*i + (lane >> 5), i = 0, 1
    const int rc4 = (lane & 31) * 4, rrow = 4 * wave + (lane >> 5);
    const ColInfo rcol = decompose(n0 + rc4, N, P, d.O2, d.O2);
    const float *rsrc = rcol.valid ? d.x + (size_t)rcol.b * d.x_bstride + rcol.p + (size_t)rrow * P : d.sink + 256;
    const size_t r_row2 = rcol.valid ? (size_t)2 * P : 0, r_step = rcol.valid ? (size_t)BK * P : 0;
#define MI_BRAW_DMA(kt, rs)                                                                                         \
    do {                                                                                                            \
        float *dst = braw + (rs) * (BK * BN) + (4 * wave) * BN;                                                     \
        const float *g = rsrc + (size_t)(kt) * r_step;                                                              \
        __builtin_amdgcn_global_load_lds((gvoid_t *)g, (lvoid_t *)dst, 16, 0, 0);                                   \
        __builtin_amdgcn_global_load_lds((gvoid_t *)(g + r_row2), (lvoid_t *)(dst + 2 * BN), 16, 0, 0);             \
    } while (0)
#define MI_BRAW_READ(rs)                                                                                            \
    do {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) breg[j] = braw[(rs) * (BK * BN) + (8 * bh + j) * BN + bn];     \
    } while (0)

#define MI_B_LOAD(kt)                                                                                               \
    do {                                                                                                            \
        if (PLAIN) {                                                                                                \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) breg[j] = bp[(size_t)(kt) * b_step + j * b_row];           \
        } else {                                                                                                    \
            mi_ktab_entry ke[8];                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) ke[j] = d.ktab[(kt) * BK + 8 * bh + j];                    \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                          \
                bool ok;                                                                                            \
                const float v = gather_b(d, ke[j], xcol, i1b, i2b, lc.valid, ok);                                   \
                breg[j] = ok ? v : 0.f;                                                                             \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)

#define MI_B_STORE(stage)                                                                                           \
    do {                                                                                                            \
        unsigned ph[4], pm[4], pl[4];                                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) split3(breg[2 * j], breg[2 * j + 1], ph[j], pm[j], pl[j]);    \
        unsigned char *bs = smem + (stage) * STAGE + A_BYTES + (bh * BN + bn) * 16;                                 \
        *reinterpret_cast<uint4 *>(bs) = make_uint4(ph[0], ph[1], ph[2], ph[3]);                                    \
        *reinterpret_cast<uint4 *>(bs + 2 * BN * 16) = make_uint4(pm[0], pm[1], pm[2], pm[3]);                      \
        *reinterpret_cast<uint4 *>(bs + 4 * BN * 16) = make_uint4(pl[0], pl[1], pl[2], pl[3]);                      \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    // the six products, smallest terms first
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    MI_A_DMA(0, 0);
    if constexpr (PLAIN) {
        MI_BRAW_DMA(0, 0);
        if (nk > 1) MI_BRAW_DMA(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MI_BRAW_READ(0);
    } else {
        MI_B_LOAD(0);
    }
    MI_B_STORE(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    bf16x8 keepa[TM][3], keepb[TN][3];       // MI_X6_ABL & 128 only
    for (int kt = 0; kt < nk; ++kt) {
        // All fragment reads of this K step are ISSUED before the loads of the next one: hipcc orders every ds_read
        // behind pending LDS-DMA with s_waitcnt vmcnt(0) (it cannot tell the ring stages apart), which would
        // otherwise make each step wait for the loads it has just issued.
        const unsigned char *As = smem + cur * STAGE, *Bs = As + A_BYTES;
        bf16x8 af[TM][3], bf[TN][3];
        if ((MI_X6_ABL & 128) && kt > 0) {   // fragments fetched in the first step only: 24 MFMAs per step, (almost) no ds_read_b128
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < TM; ++a) af[a][p] = keepa[a][p];
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[b][p] = keepb[b][p];
            }
        } else
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[a][p] = *reinterpret_cast<const bf16x8 *>(As + (((p * 2 + lh) * BM) + (wm * TM + a) * 32 + li) * 16);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bf[b][p] = *reinterpret_cast<const bf16x8 *>(Bs + (((p * 2 + lh) * BN) + (wn * TN + b) * 32 + li) * 16);
        }
        if (MI_X6_ABL & 128) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < TM; ++a) keepa[a][p] = af[a][p];
#pragma unroll
                for (int b = 0; b < TN; ++b) keepb[b][p] = bf[b][p];
            }
        }
        if (MI_X6_ABL & 64) {                 // all fragment reads stay alive although only one MFMA consumes them
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < TM; ++a) { u32x4 z = __builtin_bit_cast(u32x4, af[a][p]); asm volatile("" : "+v"(z)); af[a][p] = __builtin_bit_cast(bf16x8, z); }
#pragma unroll
                for (int b = 0; b < TN; ++b) { u32x4 z = __builtin_bit_cast(u32x4, bf[b][p]); asm volatile("" : "+v"(z)); bf[b][p] = __builtin_bit_cast(bf16x8, z); }
            }
        }
        if (MI_X6_ABL & 8) {                  // same instruction stream, all-zero operands (minimal switching power)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < TM; ++a) { u32x4 z = {0u, 0u, 0u, 0u}; asm volatile("" : "+v"(z)); af[a][p] = __builtin_bit_cast(bf16x8, z); }
#pragma unroll
                for (int b = 0; b < TN; ++b) { u32x4 z = {0u, 0u, 0u, 0u}; asm volatile("" : "+v"(z)); bf[b][p] = __builtin_bit_cast(bf16x8, z); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            MI_A_DMA(kt + 1, cur ^ 1);
            if constexpr (PLAIN) {
                if (kt + 2 < nk) MI_BRAW_DMA(kt + 2, kt & 1);   // raw stage kt & 1 was consumed during step kt - 1
            } else {
                MI_B_LOAD(kt + 1);
            }
        }
        if constexpr (PLAIN) {
            // landed and fenced by the barrier that ended step kt - 1.  Unconditional (the last step re-splits a stale
            // tile into the unused stage) so that split and MFMAs share one basic block and can be interleaved.
            MI_BRAW_READ((kt + 1) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    if (!(MI_X6_ABL & 1) || (q == 0 && a == 0 && b == 0)) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA[q]], bf[b][PB[q]], acc[a][b], 0, 0, 0);
                        if (MI_X6_ABL & 16) asm volatile("s_nop 15\n\ts_nop 15");                     // gap after every MFMA
                        if ((MI_X6_ABL & 32) && a == TM - 1 && b == TN - 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");   // gap after each group
                    }
        if (PLAIN || kt + 1 < nk) MI_B_STORE(cur ^ 1);
        if constexpr (PLAIN) {
            // the split of the next activation tile (~70 VALU) issues in the shadow of this step's MFMAs
#pragma unroll
            for (int i = 0; i < 6 * TM * TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (72 + 6 * TM * TN - 1) / (6 * TM * TN), 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the next A image / raw tile has landed
        __syncthreads();
        cur ^= 1;
    }
#undef MI_A_DMA
#undef MI_B_LOAD
#undef MI_BRAW_DMA
#undef MI_BRAW_READ
#undef MI_B_STORE

    if (MI_X6_ABL & 4) {                     // no epilogue: one store per thread keeps the loop alive
        float s = 0.f;
        for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
        d.sink[threadIdx.x] = s;
        return;
    }
    conv_epilogue<TM, TN, EPI, LFLAGS>(d, acc, m0, n0, wm, wn, N, P, o2v);
}

template <int WM, int WN, int TM, int TN, int EPI, int LFLAGS, bool PLAIN>
static int launch_cfg_x6(const mi_conv_desc &d, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    const int64_t N64 = (int64_t)d.B * d.O1 * d.O2;
    MI_REQUIRE(N64 < (1ll << 31) - 256, "conv: too many output positions (%lld)", (long long)N64);
    MI_REQUIRE(d.Mpad % BM == 0, "conv: Mpad %d not a multiple of the %d-row tile", d.Mpad, BM);
    const int N = (int)N64, MT = d.Mpad / BM, NT = ceil_div(N, BN);
    const int Gm = pick_m_groups(MT, (size_t)d.Kpad * d.Mpad * 6);
    const unsigned grid = grouped_grid(MT, NT, Gm);
    hipLaunchKernelGGL((conv_gemm_x6_kernel<WM, WN, TM, TN, EPI, LFLAGS, PLAIN>), dim3(grid), dim3(256), 0, st, d, N, MT, Gm);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int EPI, int LFLAGS, bool PLAIN>
static int launch_tile_x6(const mi_conv_desc &d, int tile, hipStream_t st) {
    switch (tile) {
        case 128: return launch_cfg_x6<2, 2, 2, 2, EPI, LFLAGS, PLAIN>(d, st);
        case 96: return launch_cfg_x6<1, 4, 3, 1, EPI, LFLAGS, PLAIN>(d, st);
        case 64: return launch_cfg_x6<1, 4, 2, 1, EPI, LFLAGS, PLAIN>(d, st);
    }
    return set_error(MI_EINVAL, "conv x6: unsupported tile_m %d", tile);
}

bool conv_x6_supported(int tile) { return tile == 128 || tile == 96 || tile == 64; }

// d has been validated by launch_conv (gemm_conv.hip), which also decided `plain`
int launch_conv_x6(const mi_conv_desc &d, int tile, bool plain, hipStream_t st) {
    g_last_conv_route = 4;
    MI_REQUIRE(d.wx && ((uintptr_t)d.wx & 15) == 0, "conv x6: split weight image missing or misaligned");
#define MI_DISPATCH(E)                                              \
    case E: return plain ? launch_tile_x6<E, 0, true>(d, tile, st) : launch_tile_x6<E, 0, false>(d, tile, st)
#define MI_LINEAR(F)                                                \
    case F: return plain ? launch_tile_x6<MI_EPI_LINEAR, F, true>(d, tile, st) : launch_tile_x6<MI_EPI_LINEAR, F, false>(d, tile, st)
    if (d.epi == MI_EPI_LINEAR) {
        switch (d.flags & (MI_FLAG_GELU | MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_LN)) {
            MI_LINEAR(0);
            MI_LINEAR(MI_FLAG_GELU);
            MI_LINEAR(MI_FLAG_RES);
            MI_LINEAR(MI_FLAG_SCALE | MI_FLAG_RES);
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

}  // namespace mi
