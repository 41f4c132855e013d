// LSTM recurrence of the BLSTM blocks of Hybrid Demucs v3 as ONE persistent launch per (layer, batch of sequence tiles)
// (reference: demucs/demucs.py:20-67 -> nn.LSTM, zero initial state, gate order i, f, g, o).
//
// Round 3 ran one launch per time step (hkernels.hip lstm_step_kernel: 1 600 dependent launches per forward, 5-8 us each: every
// launch re-read its slice of the recurrent matrix and paid a kernel boundary).  Here a launch covers all W steps:
//   * group = (direction, tile of 16 sequences); its `members` workgroups each own 4 MT hidden units (16 MT gate rows) and keep
//     their slice of W_hh in REGISTERS for the whole sequence (wave w holds the k-quarter [w H/4, (w + 1) H/4) of its rows as the
//     A operands of v_mfma_f32_16x16x4_f32: exact float32 products, as before);
//   * per step a member needs the group's whole previous hidden state h[H][16]: the members exchange it through global memory as
//     8-byte {tag = step + 1, value} granules, each written by ONE agent-scope (sc1, write-through) store and polled with agent-scope
//     loads -- the data is the flag, so no fence and no ordering between granules is needed (cdna_hip_programming.md Guideline 16,
//     form R2; price list row "allgather").  Two granule buffers alternate by step parity: a member overwrites buffer s % 2 only
//     after it has consumed every granule of step s - 1, which no member publishes before it has read all of step s - 2;
//   * lane (k4, n) of wave w loads h[w H/4 + 4 i + k4][n] straight into the B operand register of k step i: no LDS staging; the four
//     k-quarter partial blocks are summed through LDS by the gate threads (one workgroup barrier per step, double-buffered),
//     which keep the cell state in a register, publish h, and write it to the layer output.
// Correctness does not depend on where workgroups run; residency of the whole grid is required (grid <= 256 workgroups at <= 256
// VGPRs: two fit a CU), every spin is bounded by the real-time clock, and a time-out raises a flag every workgroup sees -- the
// waves then fall through the remaining steps without waiting (no early exit: barriers stay matched) and the host reports it.
#include "common.h"
#include "kernels.h"

namespace mi {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int kSeqTile = 16;
constexpr u64 kSpinTicks = 30000000ull;      // 0.3 s of the 100 MHz real-time counter per wait

__device__ __forceinline__ u64 ld_granule(const u64 *p) {
    return __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dwordx2 sc1
}
__device__ __forceinline__ void st_granule(u64 *p, u64 v) {
    __hip_atomic_store((gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // ONE global_store_dwordx2 sc1
}

// grid: 1-D, block L -> group L % G (G = 2 directions x tiles), member L / G: consecutive blocks (dealt round-robin over the XCDs)
// belong to different groups, so with G % 8 == 0 a group's members share an XCD (speed only).
// gx (N, 2, 4H, W); whh: pack_lstm_whh order; out (N, 2H, W); hx: 2 x G x H x 16 granules, zeroed by the launcher; ctl[0]: abort flag
// (zeroed by the launcher), ctl_host: the same flag in pinned host memory (sticky: read by the host without synchronising).
template <int H, int MT>
__global__ __launch_bounds__(256, 2) void lstm_persist_kernel(const float *__restrict__ gx, const float *__restrict__ whh, float *__restrict__ out,
                                                              u64 *hx, unsigned *ctl, unsigned *ctl_host, int N, int W, int n0_base, int G) {
    constexpr int KW = H / 4, NK = KW / 4, U = 4 * MT, R = 16 * MT;
    __shared__ float part[2][4][R][kSeqTile + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = blockIdx.x % G, member = blockIdx.x / G;
    const int dir = grp & 1, tile = grp >> 1;
    const int n0 = n0_base + tile * kSeqTile;
    // ---- this wave's slice of W_hh: A operands of every k step, resident for the whole sequence ---------------------------------
    float a[MT][NK];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int jb = member * MT + mt;
        const float4 *wp = (const float4 *)(whh + (((size_t)dir * (H / 4) + jb) * 4 + wave) * NK * 64) + lane;
#pragma unroll
        for (int q = 0; q < NK / 4; ++q) {
            const float4 v = wp[q * 64];
            a[mt][4 * q] = v.x; a[mt][4 * q + 1] = v.y; a[mt][4 * q + 2] = v.z; a[mt][4 * q + 3] = v.w;
        }
    }
    // ---- gate threads: thread p < 16 U owns (unit ul = p / 16, sequence p % 16) ------------------------------------------------------
    const bool gate_thread = tid < kSeqTile * U;
    const int gnl = tid & 15, gul = tid >> 4, gn = n0 + gnl, gj = member * U + gul;
    const int gmt = gul >> 2, guo = gul & 3;
    const bool gvalid = gate_thread && gn < N;
    float cstate = 0.f;
    const float *gxp = gx + (((size_t)(gvalid ? gn : 0) * 2 + dir) * 4 * H + gj) * W;
    float *outp = out + ((size_t)(gvalid ? gn : 0) * 2 * H + dir * H + gj) * W;
    u64 *hx_g = hx + (size_t)grp * H * kSeqTile;                       // + buffer * G * H * 16
    const size_t hx_buf = (size_t)G * H * kSeqTile;
    // this lane's B operand granules: h[wave * KW + 4 i + (lane >> 4)][lane & 15]
    const int poll_off = (wave * KW + (lane >> 4)) * kSeqTile + (lane & 15);
    bool dead = false;
    float pi = 0.f, pf = 0.f, pg = 0.f, po = 0.f;
    {
        const int t = dir ? W - 1 : 0;
        if (gvalid) { pi = gxp[t]; pf = gxp[(size_t)H * W + t]; pg = gxp[(size_t)2 * H * W + t]; po = gxp[(size_t)3 * H * W + t]; }
    }
    for (int s = 0; s < W; ++s) {
        const int t = dir ? W - 1 - s : s;
        v4f acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            // ---- gather the group's h_{s-1}: every granule must carry tag s ---------------------------------------------------------
            const u64 *src = hx_g + (size_t)((s - 1) & 1) * hx_buf + poll_off;
            u64 x[NK];
            bool ok = dead;
            if (!dead) {
                const u64 t0 = __builtin_amdgcn_s_memrealtime();
                for (unsigned spins = 0;; ++spins) {
                    ok = true;
#pragma unroll
                    for (int i = 0; i < NK; ++i) {
                        x[i] = ld_granule(src + (size_t)4 * i * kSeqTile);
                        ok &= (unsigned)(x[i] >> 32) == (unsigned)s;
                    }
                    if (__all(ok)) break;
                    if ((spins & 31) == 31) {           // wave-uniform: the abort flag, then the clock
                        const unsigned ab = __hip_atomic_load((gu32 *)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const bool late = __builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks;
                        if (__any(ab != 0u || late)) {
                            if (late && lane == 0) {
                                __hip_atomic_store((gu32 *)ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(ctl_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            dead = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (!dead) {
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    const float b = __uint_as_float((unsigned)x[i]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][i], b, acc[mt], 0, 0, 0);
                }
            }
        }
        // D[row 4 (lane / 16) + r][col lane % 16]; row = gate * 4 + unit offset inside the M tile
        float(*pp)[R][kSeqTile + 1] = part[s & 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[wave][mt * 16 + 4 * (lane >> 4) + r][lane & 15] = acc[mt][r];
        __syncthreads();
        if (gate_thread) {
            float s4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = gmt * 16 + g * 4 + guo;
                s4[g] = (pp[0][row][gnl] + pp[1][row][gnl]) + (pp[2][row][gnl] + pp[3][row][gnl]);
            }
            const float ai = pi + s4[0], af = pf + s4[1], ag = pg + s4[2], ao = po + s4[3];
            const float c = sigmoid_f(af) * cstate + sigmoid_f(ai) * tanhf(ag);
            const float h = sigmoid_f(ao) * tanhf(c);
            cstate = c;
            // publish: granule [unit gj][sequence gnl] of buffer s % 2, tag s + 1 (a wave's 64 lanes cover 4 units x 16 sequences = 512 B)
            st_granule(hx_g + (size_t)(s & 1) * hx_buf + (size_t)gj * kSeqTile + gnl, ((u64)(unsigned)(s + 1) << 32) | __float_as_uint(h));
            if (gvalid) {
                outp[t] = h;
                if (s + 1 < W) {                     // next step's pre-activations: in flight while the group exchanges h
                    const int tn = dir ? t - 1 : t + 1;
                    pi = gxp[tn]; pf = gxp[(size_t)H * W + tn]; pg = gxp[(size_t)2 * H * W + tn]; po = gxp[(size_t)3 * H * W + tn];
                }
            }
        }
    }
}

// tiles per launch for (H, MT): 2 directions x tiles x H / (4 MT) workgroups <= kMaxGrid
constexpr int kMaxGrid = 256;
static int tiles_per_launch(int H, int MT) { return std::max(1, kMaxGrid / (2 * (H / (4 * MT)))); }

size_t lstm_persist_scratch_bytes() {
    // granules: 2 buffers x G x H x 16 x 8 B with G x H / (4 MT) <= 256 and MT <= 3 -> at most 2 x 256 x 12 x 16 x 8 B; + control words
    return (size_t)2 * kMaxGrid * 12 * kSeqTile * sizeof(u64) + 256;
}

template <int H, int MT>
static int launch_one(const float *gx, const float *whh, float *out, u64 *hx, unsigned *ctl, unsigned *ctl_host, int N, int W, int n0, int tiles,
                      hipStream_t st) {
    const int G = 2 * tiles, members = H / (4 * MT);
    hipLaunchKernelGGL((lstm_persist_kernel<H, MT>), dim3(G * members), dim3(256), 0, st, gx, whh, out, hx, ctl, ctl_host, N, W, n0, G);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// scratch: lstm_persist_scratch_bytes() of device memory (granules first: the block the per-launch memset zeroes starts at the
// allocation's start and is a multiple of 16 bytes); ctl_host: one unsigned in pinned host memory, set (sticky) on a time-out.
int launch_lstm_persist(const float *gx, const float *whh, int N, int H, int W, float *out, void *scratch, unsigned *ctl_host, hipStream_t st) {
    MI_REQUIRE(H == 192 || H == 384, "lstm: hidden size %d not instantiated", H);
    MI_REQUIRE(N >= 1 && W >= 1, "lstm: empty problem");
    const int tiles = ceil_div(N, kSeqTile);
    // fewest hidden units per workgroup (shortest step) whose grid still fits one launch: MT = 1 for the short tail chunks, 3 for
    // a batch of 44 s chunks; beyond that the sequence tiles (they are independent) go in several launches of MT = 3
    int MT = 3;
    for (int c = 1; c <= 3; ++c)
        if (2 * tiles * (H / (4 * c)) <= kMaxGrid) { MT = c; break; }
    const int per = tiles_per_launch(H, MT);
    u64 *hx = (u64 *)scratch;
    unsigned *ctl = (unsigned *)((char *)scratch + lstm_persist_scratch_bytes() - 256);
    for (int t0 = 0; t0 < tiles; t0 += per) {
        const int nt = std::min(per, tiles - t0), G = 2 * nt;
        const size_t gran_bytes = (size_t)2 * G * H * kSeqTile * sizeof(u64);
        MI_HIP(hipMemsetAsync(hx, 0, gran_bytes, st));
        MI_HIP(hipMemsetAsync(ctl, 0, 16, st));
        const int n0 = t0 * kSeqTile;
        int r;
        if (H == 192) r = MT == 1 ? launch_one<192, 1>(gx, whh, out, hx, ctl, ctl_host, N, W, n0, nt, st)
                          : MT == 2 ? launch_one<192, 2>(gx, whh, out, hx, ctl, ctl_host, N, W, n0, nt, st)
                                    : launch_one<192, 3>(gx, whh, out, hx, ctl, ctl_host, N, W, n0, nt, st);
        else r = MT == 1 ? launch_one<384, 1>(gx, whh, out, hx, ctl, ctl_host, N, W, n0, nt, st)
                 : MT == 2 ? launch_one<384, 2>(gx, whh, out, hx, ctl, ctl_host, N, W, n0, nt, st)
                           : launch_one<384, 3>(gx, whh, out, hx, ctl, ctl_host, N, W, n0, nt, st);
        MI_TRY(r);
    }
    return MI_OK;
}

}  // namespace mi
