// Implicit-GEMM convolution on gfx950 fp32 MFMA (see gemm_conv.h).
//
// Workgroup = 256 threads = 4 waves; block tile BM x 128 (BM = 32, 64, 96 or 128), K step 16,
// LDS double buffer, global->register prefetch of tile k+1 while tile k feeds the matrix cores.
// Both operands sit in LDS k-major ([16][BM], [16][128]) so that a 32x32x2 MFMA fragment
// (lane l: row/col l&31, k = l>>5) is one conflict-free ds_read_b32 per operand.
// Arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain): the <=1e-4 parity
// target against the fp32 CPU reference leaves no room for bf16 operands.
#include <vector>

#include "gemm_tile.h"

namespace mi {

// LFLAGS: the MI_FLAG_GELU/SCALE/RES bits of a LINEAR epilogue as compile-time constants (runtime flag branches
// inside the unrolled epilogue made hipcc copy all 64 accumulators to VGPRs at once: 204 registers, 2 waves/SIMD)
template <int WM, int WN, int TM, int TN, int EPI, int LFLAGS, bool PLAIN>
__global__ __launch_bounds__(256, (TM * TN == 4 ? 3 : 4)) void conv_gemm_kernel(const mi_conv_desc d, const int N, const int MT, const int Gm) {
    constexpr int BM = WM * TM * 32;
    static_assert(WN * TN * 32 == BN, "block N tile is 128");
    static_assert(WM * WN == 4, "4 waves");
    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    if (!tile_of_block(MT, Gm, N, mt, nt)) return;      // grid padding (whole workgroup, before any barrier)
    const int m0 = mt * BM, n0 = nt * BN;
    const int P = d.O1 * d.O2;

    // ---- A loader: BK x BM floats as float4, thread-linear ---------------------------------------
    constexpr int A_F4 = BK * BM / 4, A_FULL = A_F4 / 256, A_REM = A_F4 % 256;
    constexpr int A_SLOTS = A_FULL + (A_REM ? 1 : 0);
    static_assert(A_SLOTS <= 2, "A tile fits two float4 per thread");
    const int aidx0 = tid < A_F4 ? tid : 0, aidx1 = (tid + 256 < A_F4) ? tid + 256 : 0;
    const int ar0 = aidx0 / (BM / 4), ac0 = (aidx0 % (BM / 4)) * 4;
    const int ar1 = aidx1 / (BM / 4), ac1 = (aidx1 % (BM / 4)) * 4;
    const bool a_on0 = tid < A_F4, a_on1 = tid + 256 < A_F4;
    const float *ap0 = d.wt + (size_t)ar0 * d.Mpad + m0 + ac0;
    const float *ap1 = d.wt + (size_t)ar1 * d.Mpad + m0 + ac1;
    const size_t a_step = (size_t)BK * d.Mpad;

    // ---- B loader ----------------------------------------------------------------------------------
    // generic: this thread owns column (tid & 127), rows khalf + 2*j;  plain: 4 columns x rows r, r + 8
    const int khalf = wave >> 1;
    const int o2v = d.o2_valid ? d.o2_valid : d.O2;
    const ColInfo lc = decompose(n0 + (PLAIN ? (tid & 31) * 4 : (tid & 127)), N, P, d.O2, PLAIN ? d.O2 : o2v);
    const int i1b = lc.o1 * d.S1, i2b = lc.o2 * d.S2;
    const float *xcol = d.x + (size_t)lc.b * d.x_bstride + (PLAIN ? (size_t)lc.p : (size_t)i1b * (d.x_ld ? d.x_ld : d.D2) + i2b);
    const int prow = tid >> 5;                                   // plain: first row of this thread
    const float *bp0 = lc.valid ? xcol + (size_t)prow * P : d.x; // plain row pointers (channel stride = P)
    const float *bp1 = lc.valid ? xcol + (size_t)(prow + 8) * P : d.x;
    const size_t b_step = lc.valid ? (size_t)BK * P : 0;

    float4 areg0 = make_float4(0.f, 0.f, 0.f, 0.f), areg1 = areg0, bq0 = areg0, bq1 = areg0;
    float breg[8];
    bool bok[8];

#define MI_LOAD_TILE(kt)                                                                              \
    do {                                                                                              \
        if (PLAIN) {                                                                                  \
            bq0 = *reinterpret_cast<const float4 *>(bp0 + (size_t)(kt) * b_step);                     \
            bq1 = *reinterpret_cast<const float4 *>(bp1 + (size_t)(kt) * b_step);                     \
        } else {                                                                                      \
            mi_ktab_entry ke[8];                                                                      \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) ke[j] = d.ktab[(kt) * BK + khalf + 2 * j];   \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                              \
                breg[j] = gather_b(d, ke[j], xcol, i1b, i2b, lc.valid, bok[j]);                        \
        }                                                                                             \
        if (A_SLOTS >= 1) areg0 = *reinterpret_cast<const float4 *>(ap0 + (size_t)(kt) * a_step);     \
        if (A_SLOTS >= 2) areg1 = *reinterpret_cast<const float4 *>(ap1 + (size_t)(kt) * a_step);     \
    } while (0)

#define MI_STORE_TILE(buf)                                                                            \
    do {                                                                                              \
        if (PLAIN) {                                                                                  \
            if (!lc.valid) { bq0 = make_float4(0.f, 0.f, 0.f, 0.f); bq1 = bq0; }                      \
            *reinterpret_cast<float4 *>(&Bs[buf][prow][(tid & 31) * 4]) = bq0;                        \
            *reinterpret_cast<float4 *>(&Bs[buf][prow + 8][(tid & 31) * 4]) = bq1;                    \
        } else {                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                              \
                Bs[buf][khalf + 2 * j][tid & 127] = bok[j] ? breg[j] : 0.f;                            \
        }                                                                                             \
        if (A_SLOTS >= 1 && a_on0) *reinterpret_cast<float4 *>(&As[buf][ar0][ac0]) = areg0;           \
        if (A_SLOTS >= 2 && a_on1) *reinterpret_cast<float4 *>(&As[buf][ar1][ac1]) = areg1;           \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = d.Kpad / BK;
    const int li = lane & 31, lh = lane >> 5;
    MI_LOAD_TILE(0);
    MI_STORE_TILE(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) MI_LOAD_TILE(kt + 1);
        // fragments double-buffered in registers: the reads of step s+1 are in flight under the MFMAs of step s
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = As[cur][lh][(wm * TM + a) * 32 + li];
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = Bs[cur][lh][(wn * TN + b) * 32 + li];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) {
#pragma unroll
                for (int a = 0; a < TM; ++a) af[(s + 1) & 1][a] = As[cur][2 * (s + 1) + lh][(wm * TM + a) * 32 + li];
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[(s + 1) & 1][b] = Bs[cur][2 * (s + 1) + lh][(wn * TN + b) * 32 + li];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][a], bf[s & 1][b], acc[a][b], 0, 0, 0);
        }
        // pin the software pipeline: the LDS reads of step s+1 issue BEFORE the MFMAs of step s, so their
        // latency hides under the matrix pipe (hipcc otherwise sinks each read next to its use and waits)
        constexpr int kReads = (TM == 2 ? 1 : TM) + (TN == 2 ? 1 : TN);    // ds_read2_b32 pairs two 32-apart tiles
        __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        if (kt + 1 < nk) MI_STORE_TILE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef MI_LOAD_TILE
#undef MI_STORE_TILE

    conv_epilogue<TM, TN, EPI, LFLAGS>(d, acc, m0, n0, wm, wn, N, P, o2v);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA main loop for the plain (1x1 / linear) 128 x 128 tile: both operand tiles go global -> LDS with
// global_load_lds_dwordx4 (no staging registers, no ds_write), a 3-stage LDS ring keeps TWO K tiles in flight
// behind a counted s_waitcnt vmcnt(4) and ONE raw s_barrier per K tile (cdna_hip_programming.md: "Pipelining
// across barriers").  Each wave-instruction lands 1 KiB = two 128-float rows of a [16][128] tile image.
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <int EPI, int LFLAGS>
__global__ __launch_bounds__(256, 3) void conv_gemm_dma_kernel(const mi_conv_desc d, const int N, const int MT, const int Gm) {
    constexpr int TM = 2, TN = 2, WN = 2, BM = 128;
    constexpr int SS = BK * (BM + BN);                       // floats per stage: A image then B image
    __shared__ __attribute__((aligned(16))) float smem[3 * SS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    if (!tile_of_block(MT, Gm, N, mt, nt)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int P = d.O1 * d.O2;
    const int o2v = d.o2_valid ? d.o2_valid : d.O2;

    // this lane's share of every tile: rows 4*wave + 2*j + (lane >> 5), j = 0, 1; 16 bytes at column 4*(lane & 31)
    const int r0 = 4 * wave + (lane >> 5), c4 = (lane & 31) * 4;
    const float *asrc = d.wt + (size_t)r0 * d.Mpad + m0 + c4;
    const size_t a_row2 = (size_t)2 * d.Mpad, a_step = (size_t)BK * d.Mpad;
    const ColInfo lc = decompose(n0 + c4, N, P, d.O2, d.O2);
    const float *bsrc = lc.valid ? d.x + (size_t)lc.b * d.x_bstride + lc.p + (size_t)r0 * P : d.sink + 256;
    const size_t b_row2 = lc.valid ? (size_t)2 * P : 0, b_step = lc.valid ? (size_t)BK * P : 0;

#define MI_DMA_TILE(kt, stage)                                                                                  \
    do {                                                                                                        \
        float *sa = smem + (stage) * SS + (4 * wave) * BM, *sb = smem + (stage) * SS + BK * BM + (4 * wave) * BN; \
        const float *ga = asrc + (size_t)(kt) * a_step, *gb = bsrc + (size_t)(kt) * b_step;                      \
        __builtin_amdgcn_global_load_lds((gvoid_t *)ga, (lvoid_t *)sa, 16, 0, 0);                                \
        __builtin_amdgcn_global_load_lds((gvoid_t *)(ga + a_row2), (lvoid_t *)(sa + 2 * BM), 16, 0, 0);          \
        __builtin_amdgcn_global_load_lds((gvoid_t *)gb, (lvoid_t *)sb, 16, 0, 0);                                \
        __builtin_amdgcn_global_load_lds((gvoid_t *)(gb + b_row2), (lvoid_t *)(sb + 2 * BN), 16, 0, 0);          \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = d.Kpad / BK;
    const int li = lane & 31, lh = lane >> 5;
    MI_DMA_TILE(0, 0);
    if (nk > 1) MI_DMA_TILE(1, 1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once all but this wave's newest tile (4 instructions) are done -- for every wave
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // stage (kt+2)%3 == (kt-1)%3 was last read in the previous iteration, which every wave has finished
        if (kt + 2 < nk) MI_DMA_TILE(kt + 2, stage == 0 ? 2 : stage - 1);
        const float *As = smem + stage * SS, *Bs = As + BK * BM;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = As[lh * BM + (wm * TM + a) * 32 + li];
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = Bs[lh * BN + (wn * TN + b) * 32 + li];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) {
#pragma unroll
                for (int a = 0; a < TM; ++a) af[(s + 1) & 1][a] = As[(2 * (s + 1) + lh) * BM + (wm * TM + a) * 32 + li];
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[(s + 1) & 1][b] = Bs[(2 * (s + 1) + lh) * BN + (wn * TN + b) * 32 + li];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][a], bf[s & 1][b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
#undef MI_DMA_TILE
    conv_epilogue<TM, TN, EPI, LFLAGS>(d, acc, m0, n0, wm, wn, N, P, o2v);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA main loop for STRIDE-1 k x k convolutions in float32 (the decoders' 3 x 3 / k = 3 "rewrite" convs, reference
// demucs/hdemucs.py:304-314): round 3 ran them on the table-driven gather of conv_gemm_kernel (8 dword gathers + bounds tests
// + ds_writes per thread and K step) at 0.64-0.70 of the fp32 MFMA peak against 0.74 for the DMA loop above.  Row k = (ci, tap)
// of a K step's B tile is, for 128 consecutive output positions, a run of the input row shifted by the tap's offset
// d1 * pitch + d2 -- and global_load_lds_dwordx4 takes a source that is only 4-byte aligned (tools/micro/dma_unaligned.hip), so
// the run goes global -> LDS by DMA like a plain layer's: no staging registers, no table.  What the shift drags in at the two
// ends of an input row (the neighbouring row's sample, or a pitch-padding column) is overwritten with the conv's zero padding
// in LDS by the lane that issued the transfer, after its counted wait and before the barrier: at most two ds_write_b32 per
// lane and K step.  Taps whose input ROW lies outside the frame read the zero page.  K2 = 3 (|d2| <= 1).
// Tiles: 128 x 128 (2 x 2 waves) and 96 x 128 (1 x 4 waves; the A image [16][96] is six 1-KiB transfers: waves 0, 1 issue two
// of them, waves 2, 3 one, so the counted waits are per wave).  Same 3-stage ring, one raw s_barrier per K step.
// DIL (round 4, later): step between the K2 = 3 taps of a k = 3 conv -- the DConv blocks' dilated convs C -> C / 8 (demucs.py:138: dilation
// 1 / 2, padding = dilation) with the row-statistics epilogue, on 32- and 64-row tiles (M = 24 / 48: two / four A transfers per K step).
// With DIL > 1 up to DIL samples at either end of a row come from outside it: the lanes that own the first chunk of a row and the
// chunks within eight columns of its valid end re-test their four elements per tap (a rare branch) instead of the single fix_l / fix_r.
template <int BMT, int EPI, int NT, int DIL = 1>
__global__ __launch_bounds__(256, 3) void conv_gemm_dmatap_kernel(const mi_conv_desc d, const int N, const int MT, const int Gm) {
    constexpr int BM = BMT, WM = BM == 128 ? 2 : 1, WN = 4 / WM, TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int SS = BK * (BM + BN);                       // floats per stage: A image then B image
    __shared__ __attribute__((aligned(16))) float smem[3 * SS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    if (!tile_of_block(MT, Gm, N, mt, nt)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int P = d.O1 * d.O2;
    const int o2v = d.o2_valid ? d.o2_valid : d.O2;
    constexpr int K2 = 3;                                    // NT = 9 (3 x 3) or 3 (k = 3): divisions by constants
    const int x_ld = d.x_ld ? d.x_ld : d.D2;
    const int64_t chan = (int64_t)d.D1 * x_ld;               // input channel stride
    const float *zero = d.sink + 256;

    // ---- A: this wave's transfers of every K step ------------------------------------------------------------------------------
    // BM = 128: two transfers of two 128-float rows each (rows 4 wave + 2 j + (lane >> 5)); BM = 96: transfer q moves floats
    // [256 q, 256 q + 256) of the [16][96] image, q = wave and, for waves 0 and 1, q = 4 + wave
    // BM = 64: transfer q = wave (one each); BM = 32: two transfers in all, waves 0 and 1
    constexpr int NA = BM >= 96 ? 2 : 1;                     // slots (the second one of waves 2, 3 is idle at BM = 96)
    const bool a1 = BM != 32 || wave < 2, a2 = BM == 128 || (BM == 96 && wave < 2);
    int a_row[NA], a_col[NA], a_lds[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        if (BM == 128) { a_row[j] = 4 * wave + 2 * j + (lane >> 5); a_col[j] = (lane & 31) * 4; a_lds[j] = (4 * wave + 2 * j) * BM; }
        else { const int q = j ? 4 + wave : wave, f = 256 * q + 4 * lane; a_row[j] = f / BM; a_col[j] = f % BM; a_lds[j] = 256 * q; }
    }
    // ---- B: rows 4 wave + 2 j + (lane >> 5), 16 bytes at column 4 (lane & 31) ---------------------------------------------------
    const int c4 = (lane & 31) * 4, rb0 = 4 * wave + (lane >> 5);
    const ColInfo lc = decompose(n0 + c4, N, P, d.O2, o2v);
    const float *xcol = d.x + (size_t)lc.b * d.x_bstride + (size_t)lc.o1 * x_ld + lc.o2;
    // elements of this lane's 4-column chunk that a tap with d2 = -1 / +1 reads from outside the input row [0, D2)
    const int fix_l = (lc.valid && lc.o2 == 0) ? 0 : -1;
    const int fr = d.D2 - 1 - lc.o2;
    const int fix_r = (lc.valid && fr >= 0 && fr < 4) ? fr : -1;
    const bool edge = lc.valid && (lc.o2 < 4 || lc.o2 + 8 > d.D2);       // DIL > 1: chunks that can hold an element from outside [0, D2)

#define MI_TAPDMA_TILE(kt, stage)                                                                                   \
    do {                                                                                                            \
        float *sa = smem + (stage) * SS, *sb = sa + BK * BM + (4 * wave) * BN;                                      \
        _Pragma("unroll") for (int j = 0; j < NA; ++j)                                                              \
            if (j == 0 ? a1 : a2) {                                                                                 \
                const float *ga = d.wt + (size_t)((kt) * BK + a_row[j]) * d.Mpad + m0 + a_col[j];                   \
                __builtin_amdgcn_global_load_lds((gvoid_t *)ga, (lvoid_t *)(sa + a_lds[j]), 16, 0, 0);              \
            }                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
            const int k = (kt) * BK + rb0 + 2 * j, ci = k / NT, tap = k - ci * NT, t1 = tap / K2, t2 = tap - t1 * K2; \
            const int i1 = lc.o1 + t1 - d.tap_pad1;                                                                 \
            const bool ok = lc.valid && k < d.K && (unsigned)i1 < (unsigned)d.D1;                                   \
            const float *gb = xcol + (int64_t)ci * chan + (int64_t)(t1 - d.tap_pad1) * x_ld + (t2 * DIL - d.tap_pad2); \
            __builtin_amdgcn_global_load_lds((gvoid_t *)(ok ? gb : zero), (lvoid_t *)(sb + 2 * j * BN), 16, 0, 0);  \
        }                                                                                                           \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = d.Kpad / BK;
    const int li = lane & 31, lh = lane >> 5;
    MI_TAPDMA_TILE(0, 0);
    if (nk > 1) MI_TAPDMA_TILE(1, 1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once all but this wave's newest tile (2 B transfers + its A transfers: 4, 3 or 2) are done
        if (kt + 1 < nk) {
            if (a2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (a1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {   // the conv's zero padding at the ends of the input rows, in the rows this lane transferred
            float *sb = smem + stage * SS + BK * BM + (4 * wave) * BN + (lane >> 5) * BN + c4;
            if constexpr (DIL == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = kt * BK + rb0 + 2 * j, tap = k % NT, dd2 = tap % K2 - d.tap_pad2;
                    if (dd2 < 0 && fix_l >= 0) sb[2 * j * BN + fix_l] = 0.f;
                    if (dd2 > 0 && fix_r >= 0) sb[2 * j * BN + fix_r] = 0.f;
                }
            } else if (edge) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = kt * BK + rb0 + 2 * j, tap = k % NT, dd2 = (tap % K2) * DIL - d.tap_pad2;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if ((unsigned)(lc.o2 + e + dd2) >= (unsigned)d.D2) sb[2 * j * BN + e] = 0.f;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // stage (kt+2)%3 == (kt-1)%3 was last read in the previous iteration, which every wave has finished
        if (kt + 2 < nk) MI_TAPDMA_TILE(kt + 2, stage == 0 ? 2 : stage - 1);
        const float *As = smem + stage * SS, *Bs = As + BK * BM;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = As[lh * BM + (wm * TM + a) * 32 + li];
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = Bs[lh * BN + (wn * TN + b) * 32 + li];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) {
#pragma unroll
                for (int a = 0; a < TM; ++a) af[(s + 1) & 1][a] = As[(2 * (s + 1) + lh) * BM + (wm * TM + a) * 32 + li];
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[(s + 1) & 1][b] = Bs[(2 * (s + 1) + lh) * BN + (wn * TN + b) * 32 + li];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][a], bf[s & 1][b], acc[a][b], 0, 0, 0);
        }
        constexpr int kReads = (TM == 2 ? 1 : TM) + (TN == 2 ? 1 : TN);
        __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
#undef MI_TAPDMA_TILE
    conv_epilogue<TM, TN, EPI, 0>(d, acc, m0, n0, wm, wn, N, P, o2v);
}

// float32, stride-1 k x k conv with K2 = 3 whose geometry the descriptor states (ntaps, tap_k2, tap_pad*): the DMA loop above
static bool dmatap_eligible(const mi_conv_desc &d, int tile) {
    static const bool off = getenv("MI_NO_DMA_TAP") != nullptr;
    const int ld = d.x_ld ? d.x_ld : d.D2;
    static const bool off_dconv = getenv("MI_NO_DMA_DCONV") != nullptr;      // A/B switch for the BIAS_STATS kind alone
    if (off_dconv && d.epi == MI_EPI_BIAS_STATS) return false;
    const int dil = d.tap_dil2 ? d.tap_dil2 : 1;
    // GLU: the decoders' 3 x 3 / k = 3 rewrite convs (96- / 128-row tiles); BIAS_STATS: the DConv blocks' dilated k = 3 convs (32- / 64-row)
    const bool kind = (d.epi == MI_EPI_GLU && (tile == 96 || tile == 128) && (d.ntaps == 9 || d.ntaps == 3) && dil == 1) ||
                      (d.epi == MI_EPI_BIAS_STATS && (tile == 32 || tile == 64) && d.ntaps == 3 && (dil == 1 || dil == 2) && d.flags == 0);
    return !off && !d.half && !d.wx && kind && d.tap_k2 == 3 && d.Mpad % tile == 0 &&
           d.K % d.ntaps == 0 && d.S1 == 1 && d.S2 == 1 && d.O1 == d.D1 && d.O2 == ld && ld % 4 == 0 && d.tap_pad2 == dil &&
           d.tap_pad1 == (d.ntaps / 3 - 1) / 2 && !(d.flags & (MI_FLAG_IMG | MI_FLAG_IMG4)) && d.x_bstride == (int64_t)(d.K / d.ntaps) * d.D1 * ld;
}
template <int EPI>
static int launch_dmatap(const mi_conv_desc &d, int tile, hipStream_t st) {
    const int64_t N64 = (int64_t)d.B * d.O1 * d.O2;
    MI_REQUIRE(N64 < (1ll << 31) - 256 && d.Mpad % tile == 0, "conv: DMA tap route: %lld positions, Mpad %d", (long long)N64, d.Mpad);
    const int N = (int)N64, MT = d.Mpad / tile, NT = ceil_div(N, BN);
    const unsigned grid = grouped_grid(MT, NT, 1);
    g_last_conv_route = 2;
    if constexpr (EPI == MI_EPI_BIAS_STATS) {
        const bool d2 = d.tap_dil2 == 2;
        if (tile == 64 && d2) hipLaunchKernelGGL((conv_gemm_dmatap_kernel<64, EPI, 3, 2>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
        else if (tile == 64) hipLaunchKernelGGL((conv_gemm_dmatap_kernel<64, EPI, 3, 1>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
        else if (d2) hipLaunchKernelGGL((conv_gemm_dmatap_kernel<32, EPI, 3, 2>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
        else hipLaunchKernelGGL((conv_gemm_dmatap_kernel<32, EPI, 3, 1>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
    } else if (tile == 128 && d.ntaps == 9) hipLaunchKernelGGL((conv_gemm_dmatap_kernel<128, EPI, 9>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
    else if (tile == 128) hipLaunchKernelGGL((conv_gemm_dmatap_kernel<128, EPI, 3>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
    else if (d.ntaps == 9) hipLaunchKernelGGL((conv_gemm_dmatap_kernel<96, EPI, 9>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
    else hipLaunchKernelGGL((conv_gemm_dmatap_kernel<96, EPI, 3>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA main loop for float32 layers whose taps move along ROWS only (every table entry has d2 == 0, S2 == 1): the frequency
// branch's strided encoder convs (k = 8, s = 4 along the frequency axis: row 4 o1 + j - 2), its transposed convs as two-tap GEMMs
// (rows q, q - 1) and every 1x1 layer that does not run conv_gemm_dma_kernel (GLU / GroupNorm-GLU epilogues, 96- and 64-row
// tiles).  Row k of a K step's B tile is then a run of an input row at the SAME columns as the output tile, 16-byte chunks at
// aligned positions: no shifted runs, no zero-padding fix-up, never an address outside the tensor (a tap whose row lies outside
// [0, D1), a column past the end and the K padding read the zero page).  The row offsets come from the gather table -- one
// SCALAR 64-byte load per wave and K step (its four rows' entries; lgkmcnt, so the counted vmcnt waits of the transfer ring stay
// exact) -- which keeps ONE kernel per (tile, epilogue) for all of these layers.  Summation order = conv_gemm_kernel's: results
// are bit-identical to the table-driven route (tests/test_gpu_kernels.py).  Round 3 ran these classes at 0.5-0.6 of the fp32 MFMA
// peak (8 dword gathers + bounds tests + ds_writes per thread and K step, or float4 loads staged through registers).
template <int BMT, int EPI, int LFLAGS>
__global__ __launch_bounds__(256, 3) void conv_gemm_dmarow_kernel(const mi_conv_desc d, const int N, const int MT, const int Gm) {
    constexpr int BM = BMT, WM = BM == 128 ? 2 : 1, WN = 4 / WM, TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int SS = BK * (BM + BN);                       // floats per stage: A image then B image
    __shared__ __attribute__((aligned(16))) float smem[3 * SS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    if (!tile_of_block(MT, Gm, N, mt, nt)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int P = d.O1 * d.O2;
    const int o2v = d.o2_valid ? d.o2_valid : d.O2;
    const int x_ld = d.x_ld ? d.x_ld : d.D2;
    const float *zero = d.sink + 256;

    // ---- A: the [16][BM] image of a K step is BM / 16 transfers of 1 KiB: BM = 128: two per wave (two 128-float rows each);
    //      BM = 96: six (waves 0, 1 two, waves 2, 3 one); BM = 64: one per wave.  Transfer q moves floats [256 q, 256 q + 256)
    constexpr int NA = BM == 64 ? 1 : 2;
    const bool a2 = BM == 128 || (BM == 96 && wave < 2);
    int a_row[NA], a_col[NA], a_lds[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int q = j ? 4 + wave : wave, f = 256 * q + 4 * lane;
        a_row[j] = f / BM; a_col[j] = f % BM; a_lds[j] = 256 * q;
    }
    // ---- B: rows 4 wave + 2 j + (lane >> 5), 16 bytes at column 4 (lane & 31) ---------------------------------------------------
    const int c4 = (lane & 31) * 4, lh = lane >> 5;
    const ColInfo lc = decompose(n0 + c4, N, P, d.O2, o2v);
    const int i1b = lc.o1 * d.S1;
    const float *xcol = d.x + (size_t)lc.b * d.x_bstride + (size_t)i1b * x_ld + lc.o2;
    const int4 *ktab4 = reinterpret_cast<const int4 *>(d.ktab) + 4 * wave;       // (off, d1, d2, ci) of this wave's four rows
    // (off, d1) of rows 4 wave .. 4 wave + 3 of K step kt into SGPRs.  Inline asm: hipcc turns these uniform loads into VMEM loads
    // inside the loop (the DMA builtins count as stores that might alias the table) and then waits vmcnt(0) for them -- which
    // drains the transfer ring every K step
    typedef int v2i __attribute__((ext_vector_type(2)));
    v2i e0, e1, e2, e3;
#define MI_ROW_ENTRIES(kt)                                                                                           \
    asm volatile("s_load_dwordx2 %0, %4, 0x0\n\ts_load_dwordx2 %1, %4, 0x10\n\ts_load_dwordx2 %2, %4, 0x20\n\t"          \
                 "s_load_dwordx2 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"                                               \
                 : "=&s"(e0), "=&s"(e1), "=&s"(e2), "=&s"(e3) : "s"(ktab4 + (kt) * BK) : "memory")

#define MI_ROWDMA_TILE(kt, stage)                                                                                   \
    do {                                                                                                            \
        float *sa = smem + (stage) * SS, *sb = sa + BK * BM + (4 * wave) * BN;                                      \
        _Pragma("unroll") for (int j = 0; j < NA; ++j)                                                              \
            if (j == 0 || a2) {                                                                                     \
                const float *ga = d.wt + (size_t)((kt) * BK + a_row[j]) * d.Mpad + m0 + a_col[j];                   \
                __builtin_amdgcn_global_load_lds((gvoid_t *)ga, (lvoid_t *)(sa + a_lds[j]), 16, 0, 0);              \
            }                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
            const int off = j ? (lh ? e3.x : e2.x) : (lh ? e1.x : e0.x), dd1 = j ? (lh ? e3.y : e2.y) : (lh ? e1.y : e0.y); \
            const bool ok = lc.valid && (unsigned)(i1b + dd1) < (unsigned)d.D1;                                     \
            __builtin_amdgcn_global_load_lds((gvoid_t *)(ok ? xcol + off : zero), (lvoid_t *)(sb + 2 * j * BN), 16, 0, 0); \
        }                                                                                                           \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = d.Kpad / BK;
    const int li = lane & 31;
    MI_ROW_ENTRIES(0);
    MI_ROWDMA_TILE(0, 0);
    if (nk > 1) { MI_ROW_ENTRIES(1); MI_ROWDMA_TILE(1, 1); }
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // the table rows of K step kt + 2: their latency passes under the wait for tile kt and the barrier
        if (kt + 2 < nk) MI_ROW_ENTRIES(kt + 2);
        // tile kt has landed once all but this wave's newest tile (3 or 4 transfers) are done -- for every wave
        if (kt + 1 < nk) {
            if (BM == 64 || (BM == 96 && !a2)) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // stage (kt+2)%3 == (kt-1)%3 was last read in the previous iteration, which every wave has finished
        if (kt + 2 < nk) MI_ROWDMA_TILE(kt + 2, stage == 0 ? 2 : stage - 1);
        const float *As = smem + stage * SS, *Bs = As + BK * BM;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = As[lh * BM + (wm * TM + a) * 32 + li];
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = Bs[lh * BN + (wn * TN + b) * 32 + li];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) {
#pragma unroll
                for (int a = 0; a < TM; ++a) af[(s + 1) & 1][a] = As[(2 * (s + 1) + lh) * BM + (wm * TM + a) * 32 + li];
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[(s + 1) & 1][b] = Bs[(2 * (s + 1) + lh) * BN + (wn * TN + b) * 32 + li];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][a], bf[s & 1][b], acc[a][b], 0, 0, 0);
        }
        constexpr int kReads = (TM == 2 ? 1 : TM) + (TN == 2 ? 1 : TN);
        __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
#undef MI_ROWDMA_TILE
#undef MI_ROW_ENTRIES
    conv_epilogue<TM, TN, EPI, LFLAGS>(d, acc, m0, n0, wm, wn, N, P, o2v);
}

// float32 layer with row taps only (the caller vouches for the table: `dma_rows`; a plain layer's table is the identity): see above
static bool dmarow_eligible(const mi_conv_desc &d, int tile, bool plain) {
    static const bool off = getenv("MI_NO_DMA_ROWS") != nullptr;
    const int ld = d.x_ld ? d.x_ld : d.D2;
    const int lf = d.flags & (MI_FLAG_GELU | MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_LN | MI_FLAG_STATS | MI_FLAG_IMG | MI_FLAG_IMG4 | MI_FLAG_HEADS);
    // 1x1 + GLU / GroupNorm-GLU: 128-row tiles only -- the 96-row ones are the level-0 / 1 rewrites (K = 48 / 96: three or six K steps,
    // bound by their output), where the register-staged kernel's four workgroups per CU measured 3 % faster (476 vs 490 us)
    const bool epi_ok = (d.epi == MI_EPI_LINEAR && lf == MI_FLAG_GELU && !plain) || d.epi == MI_EPI_CONVTR ||
                        ((d.epi == MI_EPI_GLU || d.epi == MI_EPI_GN_GLU) && plain && tile == 128 && !(d.flags & (MI_FLAG_IMG | MI_FLAG_IMG4)));
    return !off && !d.half && !d.wx && d.ktab && epi_ok && (plain || d.dma_rows) && (tile == 64 || tile == 96 || tile == 128) &&
           d.Mpad % tile == 0 && d.S2 == 1 && d.O2 == ld && ld % 4 == 0 && ((uintptr_t)d.x & 3) == 0 && ((uintptr_t)d.ktab & 63) == 0 &&
           (d.epi == MI_EPI_CONVTR || tile != 64 || d.epi == MI_EPI_LINEAR);
}
template <int EPI, int LFLAGS>
static int launch_dmarow(const mi_conv_desc &d, int tile, hipStream_t st) {
    const int64_t N64 = (int64_t)d.B * d.O1 * d.O2;
    MI_REQUIRE(N64 < (1ll << 31) - 256 && d.Mpad % tile == 0, "conv: DMA row route: %lld positions, Mpad %d", (long long)N64, d.Mpad);
    const int N = (int)N64, MT = d.Mpad / tile, NT = ceil_div(N, BN);
    const unsigned grid = grouped_grid(MT, NT, 1);
    g_last_conv_route = 3;
    if (tile == 128) hipLaunchKernelGGL((conv_gemm_dmarow_kernel<128, EPI, LFLAGS>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
    else if constexpr (EPI == MI_EPI_LINEAR || EPI == MI_EPI_CONVTR) {
        if (tile == 96) hipLaunchKernelGGL((conv_gemm_dmarow_kernel<96, EPI, LFLAGS>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
        else hipLaunchKernelGGL((conv_gemm_dmarow_kernel<64, EPI, LFLAGS>), dim3(grid), dim3(256), 0, st, d, N, MT, 1);
    } else return set_error(MI_EINVAL, "conv: DMA row route has no %d-row tile for epilogue %d", tile, EPI);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int EPI, int LFLAGS, bool PLAIN>
static int launch_cfg(const mi_conv_desc &d, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    const int64_t N64 = (int64_t)d.B * d.O1 * d.O2;
    MI_REQUIRE(N64 < (1ll << 31) - 256, "conv: too many output positions (%lld)", (long long)N64);
    MI_REQUIRE(d.Mpad % BM == 0, "conv: Mpad %d not a multiple of the %d-row tile", d.Mpad, BM);
    const int N = (int)N64, MT = d.Mpad / BM, NT = ceil_div(N, BN);
    // M groups (gemm_tile.h) halve this kernel's L2 misses on the 4 MiB weight matrices (TCC hit rate 48 % -> 74 %) but
    // the misses were Infinity-Cache hits: no time is gained and every activation tile is then fetched from HBM by
    // several XCDs (+10 % HBM bytes), so the fp32 path keeps one group unless MI_MGROUPS=1.
    static const bool groups = getenv("MI_MGROUPS") != nullptr;
    const int Gm = groups ? pick_m_groups(MT, (size_t)d.Kpad * d.Mpad * 4) : 1;
    const unsigned grid = grouped_grid(MT, NT, Gm);
    if constexpr (PLAIN && BM == 128 && EPI == MI_EPI_LINEAR) {
        static const bool use_dma = getenv("MI_NO_DMA") == nullptr;
        if (use_dma) {
            g_last_conv_route = 1;
            hipLaunchKernelGGL((conv_gemm_dma_kernel<EPI, LFLAGS>), dim3(grid), dim3(256), 0, st, d, N, MT, Gm);
            MI_CHECK_LAUNCH();
            return MI_OK;
        }
    }
    g_last_conv_route = 0;
    hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, TM, TN, EPI, LFLAGS, PLAIN>), dim3(grid), dim3(256), 0, st, d, N, MT, Gm);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

template <int EPI, int LFLAGS, bool PLAIN>
static int launch_tile(const mi_conv_desc &d, int tile, hipStream_t st) {
    if constexpr (PLAIN && EPI == MI_EPI_LINEAR) {
        // Small batches: a plain linear layer whose 128-row tiles give fewer workgroups than the chip has CUs (B = 1: M = 512 and
        // 2 688 tokens -> 84) runs on 64-row tiles instead -- twice the workgroups at a lower per-workgroup rate.  MI_SMALL_TILE=0: off
        static const int small = getenv("MI_SMALL_TILE") ? atoi(getenv("MI_SMALL_TILE")) : 1;
        if (small && tile == 128 && d.Mpad % 64 == 0) {
            const int64_t wgs = (int64_t)(d.Mpad / 128) * ceil_div((int64_t)d.B * d.O1 * d.O2, BN);
            if (wgs < 200) tile = 64;
        }
    }
    switch (tile) {
        case 128: return launch_cfg<2, 2, 2, 2, EPI, LFLAGS, PLAIN>(d, st);
        case 96: return launch_cfg<1, 4, 3, 1, EPI, LFLAGS, PLAIN>(d, st);
        case 64: return launch_cfg<1, 4, 2, 1, EPI, LFLAGS, PLAIN>(d, st);
        case 32: return launch_cfg<1, 4, 1, 1, EPI, LFLAGS, PLAIN>(d, st);
    }
    return set_error(MI_EINVAL, "conv: unsupported tile_m %d", tile);
}

int conv_pick_tile(int M) {
    if (M <= 32) return 32;
    if (M <= 64) return 64;
    if (M % 128 == 0) return 128;
    if (M % 96 == 0) return 96;
    if (M <= 96) return 96;
    return 128;
}

// 256-float dump for the epilogue's out-of-range stores (one word per thread), shared by all launches
// ... followed by 64 floats that stay zero (source of the LDS-DMA loader for out-of-range columns)
static float *conv_sink() {
    static float *p = nullptr;
    if (!p) {
        if (hipMalloc((void **)&p, 320 * sizeof(float)) != hipSuccess) { p = nullptr; return nullptr; }
        (void)hipMemset(p, 0, 320 * sizeof(float));
    }
    return p;
}


// 256 bytes of zeros in device memory (the tail of the sink): source of LDS-DMA loads that fall outside a tensor
const void *conv_zero_page() {
    float *p = conv_sink();
    return p ? p + 256 : nullptr;
}

// ---------------------------------------------------------------------------------------------
// MI_X6_VERIFY=1 (debugging aid for the split-bf16 path): every x6 launch is followed, on the same stream and with no
// host synchronisation, by the fp32 kernel of the same layer into a scratch copy of the output and a comparison whose
// verdict goes to a device-side log; the log is printed when the process exits.
struct X6VerifyRec { unsigned bad, maxdiff_bits; int bmin, bmax, mmin, mmax, pmin, pmax; };
struct X6VerifyInfo { int M, K, N, epi, flags, tile, plain, B, O1, O2, o2v; long long ybs, ycs; };
static X6VerifyRec *g_vlog = nullptr;
static std::vector<X6VerifyInfo> g_vinfo;
static float *g_vscratch = nullptr;
static size_t g_vscratch_elems = 0;
constexpr int kVerifyMax = 1 << 16;

__global__ void x6_verify_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n, long long ybs, long long ycs,
                                 X6VerifyRec *rec) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = a[i], y = b[i];
        const float dlt = fabsf(x - y);
        if (dlt > 1e-3f * (1.f + fabsf(y)) || (x != x) != (y != y)) {
            atomicAdd(&rec->bad, 1u);
            atomicMax(&rec->maxdiff_bits, __float_as_uint(dlt == dlt ? dlt : 3.0e38f));
            const int bb = (int)(i / ybs), m = (int)((i % ybs) / ycs), p = (int)(i % ycs);
            atomicMin(&rec->bmin, bb); atomicMax(&rec->bmax, bb);
            atomicMin(&rec->mmin, m); atomicMax(&rec->mmax, m);
            atomicMin(&rec->pmin, p); atomicMax(&rec->pmax, p);
        }
    }
}

static void x6_verify_dump() {
    if (!g_vlog || g_vinfo.empty()) return;
    (void)hipDeviceSynchronize();
    std::vector<X6VerifyRec> h(g_vinfo.size());
    if (hipMemcpy(h.data(), g_vlog, h.size() * sizeof(X6VerifyRec), hipMemcpyDeviceToHost) != hipSuccess) return;
    size_t nbad = 0;
    for (size_t i = 0; i < h.size(); ++i) {
        if (!h[i].bad) continue;
        const X6VerifyInfo &f = g_vinfo[i];
        if (++nbad <= 40)
            fprintf(stderr, "[x6 verify] launch %zu: M %d K %d N %d epi %d flags %d tile %d plain %d B %d O1 %d O2 %d o2v %d | %u bad, max %.3e, "
                    "b %d-%d  m %d-%d  p %d-%d (col tiles %d-%d)\n", i, f.M, f.K, f.N, f.epi, f.flags, f.tile, f.plain, f.B, f.O1, f.O2, f.o2v,
                    h[i].bad, __builtin_bit_cast(float, h[i].maxdiff_bits), h[i].bmin, h[i].bmax, h[i].mmin, h[i].mmax, h[i].pmin, h[i].pmax,
                    h[i].pmin / BN, h[i].pmax / BN);
    }
    fprintf(stderr, "[x6 verify] %zu launches checked, %zu with mismatches\n", h.size(), nbad);
}

static int launch_conv_fp32_only(const mi_conv_desc &d, hipStream_t st);

static int x6_verified_launch(const mi_conv_desc &d, int tile, bool plain, hipStream_t st) {
    const bool checkable = d.epi != MI_EPI_STATS_ONLY && d.epi != MI_EPI_BIAS_STATS && d.res != d.y && (int)g_vinfo.size() < kVerifyMax;
    if (!checkable) return launch_conv_x6(d, tile, plain, st);
    if (!g_vlog) {
        MI_HIP(hipMalloc((void **)&g_vlog, kVerifyMax * sizeof(X6VerifyRec)));
        std::vector<X6VerifyRec> init(kVerifyMax, X6VerifyRec{0, 0, 1 << 30, -1, 1 << 30, -1, 1 << 30, -1});
        MI_HIP(hipMemcpy(g_vlog, init.data(), init.size() * sizeof(X6VerifyRec), hipMemcpyHostToDevice));
        atexit(x6_verify_dump);
    }
    const size_t elems = (size_t)d.B * (size_t)d.y_bstride;
    if (elems > g_vscratch_elems) {
        MI_HIP(hipStreamSynchronize(st));
        if (g_vscratch) (void)hipFree(g_vscratch);
        MI_HIP(hipMalloc((void **)&g_vscratch, elems * sizeof(float)));
        g_vscratch_elems = elems;
    }
    MI_HIP(hipMemcpyAsync(g_vscratch, d.y, elems * sizeof(float), hipMemcpyDeviceToDevice, st));     // unwritten positions compare equal
    MI_TRY(launch_conv_x6(d, tile, plain, st));
    mi_conv_desc e = d;
    e.wx = nullptr; e.y = g_vscratch;
    MI_TRY(launch_conv_fp32_only(e, st));
    const int id = (int)g_vinfo.size();
    g_vinfo.push_back(X6VerifyInfo{d.M, d.K, d.B * d.O1 * d.O2, d.epi, d.flags, tile, plain ? 1 : 0, d.B, d.O1, d.O2, d.o2_valid, d.y_bstride, d.y_cstride});
    hipLaunchKernelGGL(x6_verify_kernel, dim3(1024), dim3(256), 0, st, d.y, g_vscratch, (long long)elems, (long long)d.y_bstride, (long long)d.y_cstride, g_vlog + id);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_conv(const mi_conv_desc &din, hipStream_t st) {
    mi_conv_desc d = din;
    if (!d.sink) d.sink = conv_sink();
    MI_REQUIRE(d.sink, "conv: could not allocate the store sink");
    MI_REQUIRE((int64_t)d.Mpad * d.y_cstride < (1ll << 31), "conv: output channel stride too large for 32-bit row offsets");
    MI_REQUIRE(d.Kpad % BK == 0 && d.Kpad >= BK, "conv: Kpad %d must be a positive multiple of %d", d.Kpad, BK);
    MI_REQUIRE(d.Mpad % 4 == 0, "conv: Mpad %d must be a multiple of 4", d.Mpad);
    MI_REQUIRE(d.O2 >= 32 || d.row_mode == 0 || (d.epi != MI_EPI_BIAS_STATS && d.epi != MI_EPI_STATS_ONLY),
               "conv: statistics epilogue needs O2 >= 32");
    const int tile = d.tile_m ? d.tile_m : conv_pick_tile(d.M);
    MI_REQUIRE(d.pro == 0, "conv: the fused GroupNorm+GELU prologue was replaced by launch_gn_gelu");
    MI_REQUIRE(!(d.flags & MI_FLAG_IMG4) || (d.epi == MI_EPI_GLU && d.half && d.yh && d.yh_pq > 0 && d.yh_n >= (int64_t)d.B * d.yh_pq && d.M % 32 == 0 &&
                                             ((uintptr_t)d.yh & 15) == 0),
               "conv: MI_FLAG_IMG4 needs a half-mode GLU layer with M %% 32 == 0 and an aligned phase image of >= B * yh_pq positions per plane");
    MI_REQUIRE(!(d.flags & MI_FLAG_STATS) || (d.epi == MI_EPI_LINEAR && d.stats && d.O2 >= 32 && d.row_mode == 0),
               "conv: MI_FLAG_STATS needs a LINEAR layer, a statistics buffer and O2 >= 32");
    if (d.flags & MI_FLAG_STATS) d.wx = nullptr;             // the split-bf16 main loop is not instantiated with the statistics epilogue
    MI_REQUIRE(!(d.flags & MI_FLAG_LN) || (d.pro_stats && d.scale && d.epi == MI_EPI_LINEAR), "conv: MI_FLAG_LN needs pro_stats and scale");
    // plain fast path: a 1x1 / linear layer whose gather is the identity
    const int64_t P = (int64_t)d.O1 * d.O2;
    const bool plain = d.plain && d.K == d.Kpad && P % 4 == 0 && d.x_bstride % 4 == 0 && ((uintptr_t)d.x & 15) == 0 &&
                       d.S1 == 1 && d.S2 == 1 && d.D1 == d.O1 && (d.x_ld ? d.x_ld : d.D2) == d.O2;
    MI_REQUIRE(!(d.flags & MI_FLAG_IMG) || d.epi == MI_EPI_CONVTR ||
               (d.half && d.yh && d.epi == MI_EPI_LINEAR && d.M % 8 == 0 && d.yh_n >= (int64_t)d.B * P && ((uintptr_t)d.yh & 15) == 0) ||
               (d.half && d.yh && d.epi == MI_EPI_GLU && d.M % 32 == 0 && !(d.flags & (MI_FLAG_IMG4 | MI_FLAG_EMB)) && d.yh_n >= (int64_t)d.B * P &&
                ((uintptr_t)d.yh & 15) == 0),
               "conv: MI_FLAG_IMG needs a half-precision LINEAR layer with M %% 8 == 0 (or GLU layer with M %% 32 == 0) and an aligned "
               "output image of >= B * P columns");
    MI_REQUIRE(d.epi != MI_EPI_CONVTR || ((int64_t)d.Mpad * d.y_cstride < (1ll << 30) && (d.tr_stride == 0 || d.tr_stride == 2 || d.tr_stride == 4) &&
                                          d.M % (d.tr_stride == 2 ? 2 : 4) == 0),
               "conv: transposed conv needs stride 2 or 4, M a multiple of it and 30-bit output offsets per item");
    {   // the scatter epilogue is compiled for these flag sets (gemm_tile.h convtr_tile)
        const int f = d.flags & (MI_FLAG_GELU | MI_FLAG_RES | MI_FLAG_IMG);
        MI_REQUIRE(d.epi != MI_EPI_CONVTR || f == 0 || (d.tr_stride != 2 && (f == (MI_FLAG_GELU | MI_FLAG_RES) || f == (MI_FLAG_GELU | MI_FLAG_RES | MI_FLAG_IMG))),
                   "conv: transposed conv epilogue supports no flags, GELU|RES or GELU|RES|IMG (stride 2: no flags), got %d", d.flags);
    }
    MI_REQUIRE(!(d.flags & MI_FLAG_IMG) || d.epi != MI_EPI_CONVTR ||
               (d.half && d.yh && d.yh_n >= (int64_t)d.B * d.y_cstride && ((uintptr_t)d.yh & 15) == 0),
               "conv: MI_FLAG_IMG on a transposed conv needs a half mode and an output image of >= B * y_cstride positions");
    if (d.wtap) { MI_REQUIRE(d.half && d.xh, "conv: tap-ordered weights need a half mode and an operand-image input"); g_last_conv_route = 6; return launch_conv_tap(d, tile, st); }
    MI_REQUIRE(!(d.flags & MI_FLAG_HEADS) || (d.half && d.yh && d.epi == MI_EPI_LINEAR && d.M % 512 == 0 && d.O1 == 1 && d.yh_n >= d.O2 &&
                                              ((uintptr_t)d.yh & 15) == 0 && !(d.flags & MI_FLAG_IMG)),
               "conv: MI_FLAG_HEADS needs a half-precision LINEAR layer on tokens (O1 = 1) with M %% 512 == 0 and an aligned output");
    MI_REQUIRE(!d.xh || d.half, "conv: an operand-image input needs a half-precision layer");
    if (d.half) { g_last_conv_route = 5; return launch_conv_half(d, tile, plain, st); }
    // small batches: a k x k GLU conv whose 128-row tiles under-fill the chip (B = 1: 6 x 21 workgroups) takes 96-row tiles
    static const int small_tile = getenv("MI_SMALL_TILE") ? atoi(getenv("MI_SMALL_TILE")) : 1;
    int ktile = tile;
    if (small_tile && !plain && d.epi == MI_EPI_GLU && tile == 128 && d.Mpad % 96 == 0 &&
        (int64_t)(d.Mpad / 128) * ceil_div((int64_t)d.B * d.O1 * d.O2, BN) < 200)
        ktile = 96;
    if (!plain && dmatap_eligible(d, ktile))
        return d.epi == MI_EPI_GLU ? launch_dmatap<MI_EPI_GLU>(d, ktile, st) : launch_dmatap<MI_EPI_BIAS_STATS>(d, ktile, st);
    if (dmarow_eligible(d, tile, plain)) {
        switch (d.epi) {
            case MI_EPI_LINEAR: return launch_dmarow<MI_EPI_LINEAR, MI_FLAG_GELU>(d, tile, st);
            case MI_EPI_CONVTR: return launch_dmarow<MI_EPI_CONVTR, 0>(d, tile, st);
            case MI_EPI_GLU: return launch_dmarow<MI_EPI_GLU, 0>(d, tile, st);
            case MI_EPI_GN_GLU: return launch_dmarow<MI_EPI_GN_GLU, 0>(d, tile, st);
        }
    }
    static const int x6_mode = getenv("MI_X6_MODE") ? atoi(getenv("MI_X6_MODE")) : 0;   // bisecting: 1 plain only, 2 gather only
    static const int x6_class = getenv("MI_X6_CLASS") ? atoi(getenv("MI_X6_CLASS")) : -1;   // bisecting: one kernel class only
    if (x6_class >= 0 && x6_class != d.epi * 8 + (tile == 32 ? 0 : tile == 64 ? 1 : tile == 96 ? 2 : 3) * 2 + (plain ? 1 : 0)) d.wx = nullptr;
    if (d.wx && conv_x6_supported(tile) && x6_mode == 3) return launch_conv_x6(d, tile, false, st);      // 3: table loader for all
    static const bool x6_verify = getenv("MI_X6_VERIFY") != nullptr;
    if (d.wx && conv_x6_supported(tile) && (x6_mode == 0 || (x6_mode == 1) == plain))
        return x6_verify ? x6_verified_launch(d, tile, plain, st) : launch_conv_x6(d, tile, plain, st);
#define MI_DISPATCH(E)                                              \
    case E: return plain ? launch_tile<E, 0, true>(d, tile, st) : launch_tile<E, 0, false>(d, tile, st)
#define MI_LINEAR(F)                                                \
    case F: return plain ? launch_tile<MI_EPI_LINEAR, F, true>(d, tile, st) : launch_tile<MI_EPI_LINEAR, F, false>(d, tile, st)
    if (d.epi == MI_EPI_LINEAR) {
        switch (d.flags & (MI_FLAG_GELU | MI_FLAG_SCALE | MI_FLAG_RES | MI_FLAG_LN | MI_FLAG_STATS)) {
            MI_LINEAR(0);
            MI_LINEAR(MI_FLAG_GELU);
            MI_LINEAR(MI_FLAG_RES);
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

static int launch_conv_fp32_only(const mi_conv_desc &d, hipStream_t st) {
    mi_conv_desc e = d;
    e.wx = nullptr;
    return launch_conv(e, st);
}

}  // namespace mi
