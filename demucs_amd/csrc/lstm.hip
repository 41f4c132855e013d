// LSTM recurrence of the BLSTM blocks of Hybrid Demucs v3 as ONE persistent launch per (layer, batch of sequence tiles)
// (reference: demucs/demucs.py:20-67 -> nn.LSTM, zero initial state, gate order i, f, g, o).
//
// Round 3 ran one launch per time step (hkernels.hip lstm_step_kernel: 1 600 dependent launches per forward, 5-8 us each: every
// launch re-read its slice of the recurrent matrix and paid a kernel boundary).  Here a launch covers all W steps:
//   * group = (direction, tile of 16 sequences); its `members` workgroups each own 4 MT hidden units (16 MT gate rows) and keep
//     their slice of W_hh in REGISTERS for the whole sequence (wave w holds the k-quarter [w H/4, (w + 1) H/4) of its rows as the
//     A operands of v_mfma_f32_16x16x4_f32: exact float32 products, as before);
//   * per step a member needs the group's whole previous hidden state h[H][16].  The members exchange it through global memory as
//     SELF-VALIDATING 4-byte values: the least significant mantissa bit of every stored h carries a validity tag (lstm_tag: it
//     flips every second step), two buffers alternate by step parity, so a reader tells "h of step s" from the buffer's previous
//     occupant (step s - 2, other tag) or its zero fill without any flag, fence or ordering between values (the data is the flag:
//     cdna_hip_programming.md Guideline 16, form R2, at half the bytes of {tag, value} granules -- the first version used those and
//     measured 2.2 us per 48 KB gather, bound by the Infinity Cache's bandwidth: 256 workgroups x 48 KB per step).  A member
//     overwrites buffer s % 2 only after it has consumed every value of step s - 1, which no member publishes before it has read
//     all of step s - 2.  The tag costs h at most one ulp; hkernels.hip's step kernel applies the same rule (shared lstm_cell),
//     so both routes give identical bits;
//   * lane (k4, n) of wave w loads h[w H/4 + 4 i + k4][n] straight into the B operand register of k step i: no LDS staging; the four
//     k-quarter partial blocks are summed through LDS by the gate threads (one workgroup barrier per step, double-buffered),
//     which keep the cell state in a register, publish h, and write it to the layer output.
// PLACEMENT.  Correctness never depends on where workgroups run.  Loads are agent-scope (sc1: bypass the CU's L1, served by L2 or
// beyond).  Stores: block ids are dealt so that a group's members have equal blockIdx % 8 -- under the dispatcher's round-robin
// they then share an XCD -- and every group CHECKS that at run time (each member publishes its HW_REG_XCC_ID write-through, all
// compare): only a group that found all its members on one XCD stores h with plain stores (they stay in that XCD's L2, which is
// coherent for all its CUs: the gather is then an L2 hit); any other group stores write-through (sc1), visible chip-wide.
// Residency of the whole grid is required (<= 256 workgroups at <= 256 VGPRs: two fit a CU), every spin is bounded by the
// real-time clock, and a time-out raises a flag every workgroup sees -- the waves then fall through the remaining steps without
// waiting (no early exit: barriers stay matched) and the host reports it (hmodel.hip, mi_lstm_seq).
#include "common.h"
#include "kernels.h"

namespace mi {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int kSeqTile = 16;
constexpr u64 kSpinTicks = 30000000ull;      // 0.3 s of the 100 MHz real-time counter per wait

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) {
    return __hip_atomic_load((gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dword sc1
}
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) {
    __hip_atomic_store((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // global_store_dword sc1 (write-through)
}

// One bounded wait: `ready()` (wave-uniform result) is re-evaluated until true, the abort flag is up or the clock runs out.
// Returns false when the wave must stop waiting for good (it then falls through the remaining steps).
template <typename F>
__device__ __forceinline__ bool bounded_wait(F ready, unsigned *ctl, unsigned *ctl_host, int lane, unsigned &passes) {
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 0;; ++spins) {
        ++passes;
        if (ready()) return true;
        if ((spins & 31) == 31) {           // wave-uniform: the abort flag, then the clock
            const unsigned ab = ld_agent(ctl);
            const bool late = __builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks;
            if (__any(ab != 0u || late)) {
                if (late && lane == 0) {
                    st_agent(ctl, 1u);
                    __hip_atomic_store(ctl_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// grid: 1-D, block L -> XCD residue x = L % 8, slot q = L / 8 -> group x + 8 (q % gpx), member q / gpx: all members of a group have the
// same L % 8 (blocks b and b + 8 share an XCD under the observed round-robin dispatch; verified per group at run time, see above).
// Blocks whose group index is >= G exit at once.
// gx (N, 2, 4H, W); whh: pack_lstm_whh order; out (N, 2H, W); hx: 2 x G x H x 16 values, zeroed by the launcher; xcc: G x members
// words, zeroed by the launcher; ctl[0]: abort flag (zeroed by the launcher), ctl[2..5]: debug counters; ctl_host: the abort flag in
// pinned host memory (sticky: read by the host without synchronising).
template <int H, int MT>
__global__ __launch_bounds__(256, 2) void lstm_persist_kernel(const float *__restrict__ gx, const float *__restrict__ whh, float *__restrict__ out,
                                                              unsigned *hx, unsigned *xcc, unsigned *ctl, unsigned *ctl_host, int N, int W,
                                                              int n0_base, int G, int gpx, int force_wt) {
    constexpr int KW = H / 4, NK = KW / 4, U = 4 * MT, R = 16 * MT, MEMBERS = H / U;
    __shared__ __attribute__((aligned(16))) float part[2][R][kSeqTile][4];     // [buffer][gate row][sequence][k-quarter = wave]
    __shared__ int s_local;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xres = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = xres + 8 * (slot % gpx), member = slot / gpx;
    if (grp >= G) return;                                  // padding block of the XCD-aligned grid (whole workgroup, before any barrier)
    const int dir = grp & 1, tile = grp >> 1;
    const int n0 = n0_base + tile * kSeqTile;
    unsigned passes = 0;
    bool dead = false;
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
    // ---- where does this group run?  Every member publishes its XCD id (write-through) and reads all of them ------------------------
    if (wave == 0) {
        const unsigned my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;          // HW_REG_XCC_ID, bits [3:0]
        unsigned *tab = xcc + (size_t)grp * MEMBERS;
        if (lane == 0) st_agent(tab + member, 0x100u | my_xcc);
        const bool got = bounded_wait([&]() {
            bool ok = true;
            for (int m = lane; m < MEMBERS; m += 64) ok &= ld_agent(tab + m) != 0u;
            return (bool)__all(ok);
        }, ctl, ctl_host, lane, passes);
        bool same = got;
        if (got)
            for (int m = lane; m < MEMBERS; m += 64) same &= ld_agent(tab + m) == (0x100u | my_xcc);
        const bool all_same = __all(same);
        if (lane == 0) s_local = !got ? -1 : (all_same && !force_wt) ? 1 : 0;
    }
    __syncthreads();
    const bool plain_store = s_local == 1;                 // the whole group sits on one XCD: h stays in its L2
    dead = s_local < 0;
    // ---- gate threads: thread p < 16 U owns (unit ul = p / 16, sequence p % 16) ------------------------------------------------------
    const bool gate_thread = tid < kSeqTile * U;
    const int gnl = tid & 15, gul = tid >> 4, gn = n0 + gnl, gj = member * U + gul;
    const int gmt = gul >> 2, guo = gul & 3;
    const bool gvalid = gate_thread && gn < N;
    float cstate = 0.f;
    const float *gxp = gx + (((size_t)(gvalid ? gn : 0) * 2 + dir) * 4 * H + (gate_thread ? gj : 0)) * W;
    float *outp = out + ((size_t)(gvalid ? gn : 0) * 2 * H + dir * H + (gate_thread ? gj : 0)) * W;
    // the exchange buffers through a buffer descriptor: the gather's loads are ordinary (counted, batched) loads to the compiler
    // with the sc1 bit in `aux` -- relaxed atomic loads were serialised two by two (a wait after every pair: 12 round trips per gather)
    const unsigned hx_buf_bytes = (unsigned)G * H * kSeqTile * 4u;                         // one buffer; two alternate by step parity
    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 2 * hx_buf_bytes, 0x00020000);
    const unsigned hx_grp_bytes = (unsigned)grp * H * kSeqTile * 4u;
    // this lane's B operands: h[wave * KW + 4 i + (lane >> 4)][lane & 15]
    const unsigned poll_off = hx_grp_bytes + (unsigned)((wave * KW + (lane >> 4)) * kSeqTile + (lane & 15)) * 4u;
    const unsigned pub_off = hx_grp_bytes + (unsigned)((gate_thread ? gj : 0) * kSeqTile + gnl) * 4u;
    u64 dbg_ticks = 0;               // real-time ticks (10 ns) spent gathering h (reported by block 0, wave 1)
#ifdef MI_LSTM_STAMPS                // diagnostic build (make EXTRA=-DMI_LSTM_STAMPS): shader-clock cycles per phase of a step
    u64 ph[5] = {0, 0, 0, 0, 0};     // gather, issue + products, LDS write + barrier, gate stage, whole step
    u64 c_end = __builtin_amdgcn_s_memtime();
#define MI_STAMP(name) const u64 name = __builtin_amdgcn_s_memtime()
#else
#define MI_STAMP(name)
#endif
    // Vector-memory operations complete in issue order (vmcnt), so whatever a wave issues just before its gather delays it: the
    // layer-output store of step s - 1 and the pre-activation loads of step s + 1 are issued right AFTER the gather of step s has
    // returned -- they then have the products and the gate stage (> 1 us) to finish before the next gather is issued.
    // The pre-activations of a (unit, sequence) are W consecutive floats per gate, so with W % 4 == 0 a thread fetches FOUR steps
    // per gate with one 16-byte load and stores four outputs with one 16-byte store: a quarter of the instructions, each of which
    // costs the issuing wave ~250 cycles (64 lanes in 64 different cache lines; measured 650-1 650 cycles per step with scalar
    // accesses).  Chunk c = steps 4c .. 4c + 3 = times [4c, 4c + 3] forward, [W - 4 - 4c, W - 1 - 4c] backward (element 3 - r).
    const bool vec4 = (W & 3) == 0;
    float4 cur[4], nxt[4], hbuf = make_float4(0.f, 0.f, 0.f, 0.f), hsend = hbuf;
#pragma unroll
    for (int g = 0; g < 4; ++g) cur[g] = nxt[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    float pi = 0.f, pf = 0.f, pg = 0.f, po = 0.f;          // scalar route: pre-activations of the current step ...
    float ni = 0.f, nf = 0.f, ng = 0.f, no = 0.f;          // ... and of the next one
    float h_prev = 0.f;
    if (gvalid) {
        if (vec4) {
            const int tb = dir ? W - 4 : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) cur[g] = *reinterpret_cast<const float4 *>(gxp + (size_t)g * H * W + tb);
        } else {
            const int t = dir ? W - 1 : 0;
            pi = gxp[t]; pf = gxp[(size_t)H * W + t]; pg = gxp[(size_t)2 * H * W + t]; po = gxp[(size_t)3 * H * W + t];
        }
    }
    // the current step's element is always the FRONT of the chunk registers (x forward, w backward): after each step they rotate by
    // one element -- four selects per gate on the (wave-uniform) direction, no branch tree over the element index
    auto rotate = [&](float4 &v) {
        const float4 o = v;
        v.x = dir ? o.w : o.y; v.y = dir ? o.x : o.z; v.z = dir ? o.y : o.w; v.w = dir ? o.z : o.x;
    };
    for (int s = 0; s < W; ++s) {
        const int t = dir ? W - 1 - s : s;
        v4f acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
        unsigned x[NK];
        MI_STAMP(c0);
        if (s > 0 && !dead) {
            // ---- gather the group's h_{s-1}: every value must carry the tag of step s - 1 ------------------------------------------------
            const unsigned src_buf = ((s - 1) & 1) ? hx_buf_bytes : 0u;     // wave-uniform: the buffer instruction's scalar offset
            const unsigned want = lstm_tag(s - 1);
            const u64 t0 = __builtin_amdgcn_s_memrealtime();
            dead = !bounded_wait([&]() {
                asm volatile("" ::: "memory");             // every pass re-reads memory
#pragma unroll
                for (int i = 0; i < NK; ++i)               // all NK loads in flight together (sc1: aux = 16)
                    x[i] = __builtin_amdgcn_raw_buffer_load_b32(hx_rsrc, poll_off + (unsigned)(4 * i * kSeqTile * 4), src_buf, 16);
                __builtin_amdgcn_sched_barrier(0);         // ... before the first of them is looked at (hipcc issued them 8 at a time)
                unsigned bad = 0u;
#pragma unroll
                for (int i = 0; i < NK; ++i) bad |= x[i] ^ want;
                return (bool)__all((bad & 1u) == 0u);
            }, ctl, ctl_host, lane, passes);
            dbg_ticks += __builtin_amdgcn_s_memrealtime() - t0;
        }
        MI_STAMP(c1);
        if (gvalid) {
            if (vec4) {
                if ((s & 3) == 0) {                        // once per chunk: the finished chunk's outputs, the next chunk's pre-activations
                    if (s > 0) *reinterpret_cast<float4 *>(outp + (dir ? t + 1 : t - 4)) = hsend;
                    if (s + 4 < W) {
                        const int tb = dir ? t - 7 : t + 4;
#pragma unroll
                        for (int g = 0; g < 4; ++g) nxt[g] = *reinterpret_cast<const float4 *>(gxp + (size_t)g * H * W + tb);
                    }
                }
            } else {
                if (s > 0) outp[dir ? t + 1 : t - 1] = h_prev;
                if (s + 1 < W) {
                    const int tn = dir ? t - 1 : t + 1;
                    ni = gxp[tn]; nf = gxp[(size_t)H * W + tn]; ng = gxp[(size_t)2 * H * W + tn]; no = gxp[(size_t)3 * H * W + tn];
                }
            }
        }
        if (s > 0 && !dead) {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const float b = __uint_as_float(x[i]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][i], b, acc[mt], 0, 0, 0);
            }
        }
#ifdef MI_LSTM_STAMPS
        asm volatile("s_nop 0" : "+v"(acc[0]));        // the products are complete here
#endif
        MI_STAMP(c2);
        // D[row 4 (lane / 16) + r][col lane % 16]; row = gate * 4 + unit offset inside the M tile
        float(*pp)[kSeqTile][4] = part[s & 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[mt * 16 + 4 * (lane >> 4) + r][lane & 15][wave] = acc[mt][r];
        // LDS only: __syncthreads() would also wait (vmcnt(0)) for the layer-output store and the pre-activation loads just issued
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MI_STAMP(c3);
        if (gate_thread) {
            float s4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q = *reinterpret_cast<const float4 *>(pp[gmt * 16 + g * 4 + guo][gnl]);      // the four k-quarter partial sums
                s4[g] = (q.x + q.y) + (q.z + q.w);
            }
            if (vec4) {
                pi = dir ? cur[0].w : cur[0].x; pf = dir ? cur[1].w : cur[1].x; pg = dir ? cur[2].w : cur[2].x; po = dir ? cur[3].w : cur[3].x;
            }
            const float h = lstm_cell(pi + s4[0], pf + s4[1], pg + s4[2], po + s4[3], cstate, lstm_tag(s));
            // publish: value [unit gj][sequence gnl] of buffer s % 2 (a wave's 64 lanes cover 4 units x 16 sequences = 256 B)
            const unsigned dst_buf = (s & 1) ? hx_buf_bytes : 0u;
            if (plain_store) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), hx_rsrc, pub_off, dst_buf, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), hx_rsrc, pub_off, dst_buf, 16);       // sc1: write-through
            if (vec4) {
                // outputs enter at the back (forward) / front (backward): after four steps the chunk is in time order
                const float4 o = hbuf;
                hbuf.x = dir ? h : o.y; hbuf.y = dir ? o.x : o.z; hbuf.z = dir ? o.y : o.w; hbuf.w = dir ? o.z : h;
                if ((s & 3) == 3) {
                    hsend = hbuf;
#pragma unroll
                    for (int g = 0; g < 4; ++g) cur[g] = nxt[g];
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) rotate(cur[g]);
                }
            } else {
                h_prev = h;
                pi = ni; pf = nf; pg = ng; po = no;
            }
        }
#ifdef MI_LSTM_STAMPS
        MI_STAMP(c4);
        ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3; ph[4] += c4 - c_end;
        c_end = c4;
#endif
    }
    if (gvalid) {
        if (vec4) *reinterpret_cast<float4 *>(outp + (dir ? 0 : W - 4)) = hsend;
        else outp[dir ? 0 : W - 1] = h_prev;
    }
    if (blockIdx.x == 0 && tid == 64) {
        ctl[2] = (unsigned)dbg_ticks; ctl[3] = passes; ctl[4] = plain_store ? 1u : 0u;
#ifdef MI_LSTM_STAMPS
        for (int q = 0; q < 5; ++q) ctl[8 + q] = (unsigned)ph[q];
#endif
    }
}

constexpr int kMaxGroups = 16;     // per launch: two groups per XCD at most

static constexpr size_t kValBytes = (size_t)2 * kMaxGroups * 384 * kSeqTile * sizeof(unsigned);   // 2 buffers x 16 groups x H x 16 values
static constexpr size_t kXccBytes = (size_t)kMaxGroups * 96 * sizeof(unsigned);                    // XCD table: groups x members

size_t lstm_persist_scratch_bytes() { return kValBytes + kXccBytes + 256; }
size_t lstm_persist_ctl_offset() { return kValBytes + kXccBytes; }

template <int H, int MT>
static int launch_one(const float *gx, const float *whh, float *out, unsigned *hx, unsigned *xcc, unsigned *ctl, unsigned *ctl_host, int N, int W,
                      int n0, int G, hipStream_t st) {
    constexpr int members = H / (4 * MT);
    const int gpx = ceil_div(G, 8);
    static const int force_wt = getenv("MI_LSTM_WRITE_THROUGH") != nullptr;      // A/B: never use the same-XCD plain-store path
    hipLaunchKernelGGL((lstm_persist_kernel<H, MT>), dim3(8 * gpx * members), dim3(256), 0, st, gx, whh, out, hx, xcc, ctl, ctl_host, N, W, n0, G, gpx,
                       force_wt);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

// scratch: lstm_persist_scratch_bytes() of device memory (the blocks the per-launch memsets zero start on 16-byte boundaries
// and are multiples of 16 bytes); ctl_host: one unsigned in pinned host memory, set (sticky) on a time-out.
int launch_lstm_persist(const float *gx, const float *whh, int N, int H, int W, float *out, void *scratch, unsigned *ctl_host, hipStream_t st) {
    MI_REQUIRE(H == 192 || H == 384, "lstm: hidden size %d not instantiated", H);
    MI_REQUIRE(N >= 1 && W >= 1, "lstm: empty problem");
    const int tiles = ceil_div(N, kSeqTile);
    // A group's members share an XCD (32 CUs): H = 384 -> 32 members of 12 hidden units (MT = 3), one group per XCD and launch (8
    // groups = 64 sequences); H = 192 -> 24 members of 8 units while one group per XCD suffices, 16 members of 12 units and two
    // groups per XCD beyond (16 groups = 128 sequences per launch).  More sequence tiles (they are independent) go in further launches.
    const int g_all = 2 * tiles;
    const int MT = H == 384 ? 3 : (g_all <= 8 ? 2 : 3);
    const int g_max = H == 384 ? 8 : (MT == 2 ? 8 : kMaxGroups);
    unsigned *hx = (unsigned *)scratch;
    unsigned *xcc = (unsigned *)((char *)scratch + kValBytes);
    unsigned *ctl = (unsigned *)((char *)scratch + kValBytes + kXccBytes);
    for (int t0 = 0; t0 < tiles; t0 += g_max / 2) {
        const int nt = std::min(g_max / 2, tiles - t0), G = 2 * nt;
        MI_HIP(hipMemsetAsync(hx, 0, (size_t)2 * G * H * kSeqTile * sizeof(unsigned), st));
        MI_HIP(hipMemsetAsync(xcc, 0, kXccBytes + 32, st));                    // XCD table + the control words
        const int n0 = t0 * kSeqTile;
        int r;
        if (H == 192) r = MT == 2 ? launch_one<192, 2>(gx, whh, out, hx, xcc, ctl, ctl_host, N, W, n0, G, st)
                                  : launch_one<192, 3>(gx, whh, out, hx, xcc, ctl, ctl_host, N, W, n0, G, st);
        else r = launch_one<384, 3>(gx, whh, out, hx, xcc, ctl, ctl_host, N, W, n0, G, st);
        MI_TRY(r);
    }
    return MI_OK;
}

}  // namespace mi
