// STFT / iSTFT for the HTDemucs segment forward on gfx950.
//
// Replaces `_spec`/`_magnitude` and `_mask`/`_ispec` (reference: demucs/htdemucs.py:420-471,
// demucs/spec.py:11-47): n_fft 4096, hop 1024, periodic Hann, normalized, centre reflect padding.
//
// One 256-thread workgroup computes one 4096-point complex FFT entirely in LDS (Stockham
// autosort, three radix-16 passes, twiddles staged in LDS).  The two audio channels of a frame
// are packed as real/imaginary parts of ONE complex transform, halving the FFT count.
// A workgroup walks a run of consecutive frames (forward: stft_walk_kernel; inverse: istft_fused_kernel, which also keeps the
// overlap-add in registers) with the next frame's data prefetched under the current frame's passes.
// Layouts:   frame-major scratch  zt[b][t][4][2048]   (coalesced frame stores), then a strip
// transpose applies the per-item normalisation and writes the conv layout x[b][4][2048][T].
#include "common.h"
#include "kernels.h"

namespace mi {

constexpr int kN = 4096;
constexpr int kHop = 1024;
constexpr int kBins = 2048;
__device__ __forceinline__ int lpad(int i) { return i + (i >> 5); }   // LDS index padding (bank spread)
constexpr int kLdsN = kN + kN / 32;

// complex values as float32 pairs in adjacent registers: + and - are one v_pk_add_f32; a complex product is two v_pk_mul_f32 and
// one v_pk_add_f32 (separate IEEE multiplies and adds, as before: nothing is contracted), the multiplication by +-i of the
// radix-4 butterfly rides on the operand selectors of the add.  Written as inline asm: from C++ swizzles hipcc builds the same
// values with v_mov / v_xor around the packed instructions.  The transforms are bound by VALU issue (tools/micro/README.md).
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cmul(cf a, cf b) {
    cf p1, p2, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p1) : "v"(a), "v"(b));      // (ax bx, ax by)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p2) : "v"(a), "v"(b));      // (ay by, ay bx)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p1), "v"(p2));                     // (ax bx - ay by, ax by + ay bx)
    return r;
}
// a + i b = (ax - by, ay + bx) and a - i b = (ax + by, ay - bx)
__device__ __forceinline__ cf add_i(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf sub_i(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool INV>
__device__ __forceinline__ void dft4(cf &a, cf &b, cf &c, cf &d) {
    const cf t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    a = t0 + t2; c = t0 - t2;
    if (INV) { b = add_i(t1, t3); d = sub_i(t1, t3); }        // t1 +- (+i) t3
    else { b = sub_i(t1, t3); d = add_i(t1, t3); }            // t1 +- (-i) t3
}

// in-register 16-point DFT; result X[k1 + 4*k2] is left in u[4*k1 + k2]
template <bool INV>
__device__ __forceinline__ void dft16(cf (&u)[16]) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(u[n2], u[4 + n2], u[8 + n2], u[12 + n2]);
    // W16^m, m = n2*k1
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r2 = 0.70710678118654752440f;
    const cf w1 = {c1, INV ? s1 : -s1}, w2 = {r2, INV ? r2 : -r2}, w3 = {s1, INV ? c1 : -c1};
    const cf w4 = {0.f, INV ? 1.f : -1.f}, w6 = {-r2, INV ? r2 : -r2}, w9 = {-c1, INV ? -s1 : s1};
    u[4 * 1 + 1] = cmul(u[4 * 1 + 1], w1); u[4 * 1 + 2] = cmul(u[4 * 1 + 2], w2); u[4 * 1 + 3] = cmul(u[4 * 1 + 3], w3);
    u[4 * 2 + 1] = cmul(u[4 * 2 + 1], w2); u[4 * 2 + 2] = cmul(u[4 * 2 + 2], w4); u[4 * 2 + 3] = cmul(u[4 * 2 + 3], w6);
    u[4 * 3 + 1] = cmul(u[4 * 3 + 1], w3); u[4 * 3 + 2] = cmul(u[4 * 3 + 2], w6); u[4 * 3 + 3] = cmul(u[4 * 3 + 3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(u[4 * k1], u[4 * k1 + 1], u[4 * k1 + 2], u[4 * k1 + 3]);
}

// One Stockham pass: u[] already holds x[i + 256 r] (r = 0..15) of thread i; P = 1, 16 or 256.
// Writes the pass output to LDS re/im (natural order after the P = 256 pass).
// The twiddle half table sits in LDS with the same one-word-per-32 padding as the data (index lpad(h)): the passes read it at
// strides 16 r (P = 16) and r (P = 256) words, which on the unpadded table put the 16 / 32 lanes of an access group on one to four
// banks (8- to 16-way conflicts: SQ_LDS_BANK_CONFLICT was 56 % of the LDS cycles of the fused iSTFT); the padding spreads every
// power-of-two stride over the 32 banks.
template <bool INV, int P>
__device__ __forceinline__ void stockham_pass(cf (&u)[16], int i, float *re, float *im, const float *twr, const float *twi) {
    const int k = i & (P - 1);
    if (P > 1) {
        const int step = k * (kN / (16 * P));
#pragma unroll
        for (int r = 1; r < 16; ++r) {
            // half table: W[n + 2048] = -W[n]
            const int n = step * r, h = lpad(n & (kN / 2 - 1));
            const float sg = (n & (kN / 2)) ? -1.f : 1.f;
            const cf w = {sg * twr[h], (INV ? -sg : sg) * twi[h]};
            u[r] = cmul(u[r], w);
        }
    }
    dft16<INV>(u);
    const int j = (i - k) * 16 + k;
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const int r = k1 + 4 * k2;
            const int o = lpad(j + r * P);
            re[o] = u[4 * k1 + k2].x;
            im[o] = u[4 * k1 + k2].y;
        }
}

template <bool INV>
__device__ __forceinline__ void load_pass_input(cf (&u)[16], int i, const float *re, const float *im) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = lpad(i + 256 * r);
        u[r] = {re[o], im[o]};
    }
}

// Workgroup barrier for LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load (s_waitcnt vmcnt(0)), which
// would end the fused iSTFT's prefetch of the next frame at the first barrier of the current one
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// passes P=16 and P=256 reading from / writing to LDS (pass P=1 is done by the caller)
template <bool INV>
__device__ __forceinline__ void fft_tail(cf (&u)[16], int i, float *re, float *im, const float *twr, const float *twi) {
    lds_barrier();
    load_pass_input<INV>(u, i, re, im);
    lds_barrier();
    stockham_pass<INV, 16>(u, i, re, im, twr, twi);
    lds_barrier();
    load_pass_input<INV>(u, i, re, im);
    lds_barrier();
    stockham_pass<INV, 256>(u, i, re, im, twr, twi);
    lds_barrier();
}

constexpr int kTwLds = kN / 2 + kN / 64;       // padded half table
__device__ __forceinline__ void stage_twiddles(const float2 *tw, float *twr, float *twi, int i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float2 w = tw[i + 256 * r];
        twr[lpad(i + 256 * r)] = w.x;
        twi[lpad(i + 256 * r)] = w.y;
    }
}
#define MI_FFT_LDS __shared__ float re[kLdsN], im[kLdsN], twr[kTwLds], twi[kTwLds]

// ---------------------------------------------------------------------------------------------
// STFT: grid (T, B), block 256.  mix (B,2,L) -> zt[b][t][4][2048] + per-item (sum, sumsq) in fp64.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_frames_kernel(const float *__restrict__ mix, int L, int T, int el, int er,
                                                          const float *__restrict__ window, const float2 *__restrict__ tw,
                                                          float *__restrict__ zt, double *__restrict__ stats) {
    MI_FFT_LDS;
    __shared__ double red[8];
    const int i = threadIdx.x, t = blockIdx.x, b = blockIdx.y;
    stage_twiddles(tw, twr, twi, i);
    const float *x0 = mix + (size_t)b * 2 * L, *x1 = x0 + L;
    // frame t of the kept range is frame t+2 of th.stft: padded-signal samples [(t+2)*1024 - 2048, +4096)
    // of x1 = reflect_pad(mix, 1536, 1536 + T*1024 - L)  (htdemucs.py:433-435); the centre padding of
    // th.stft itself never reaches the kept frames.
    // x1 = pad1d(mix, (1536, 1536 + T*1024 - L), "reflect") (htdemucs.py:433-435, hdemucs.py:23-40): when the input is not
    // longer than the larger padding, pad1d first zero-pads it by (el, er) to L0 = max_pad + 1 samples and reflects the rest
    const int L0 = L + el + er, shift = 1536 - el;
    cf u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = i + 256 * r;
        int q = t * kHop + p - shift;               // index into the zero-padded signal x0 (kept frames never reach th.stft's own centre padding)
        if (q < 0) q = -q;
        if (q >= L0) q = 2 * (L0 - 1) - q;
        const int s = q - el;
        const bool in = s >= 0 && s < L;
        const float w = window[p];
        u[r] = {in ? w * x0[s] : 0.f, in ? w * x1[s] : 0.f};
    }
    __syncthreads();   // twiddles staged
    stockham_pass<false, 1>(u, i, re, im, twr, twi);
    fft_tail<false>(u, i, re, im, twr, twi);
    // split the packed transform: X0 = (Z[k] + conj Z[N-k]) / 2, X1 = (Z[k] - conj Z[N-k]) / (2i); x 1/sqrt(N)
    float *o = zt + ((size_t)b * T + t) * 4 * kBins;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = i + 256 * r;
        const int km = (kN - k) & (kN - 1);
        const float ar = re[lpad(k)], ai = im[lpad(k)], br = re[lpad(km)], bi = im[lpad(km)];
        const float sc = 0.5f / 64.0f;
        const float v0 = (ar + br) * sc, v1 = (ai - bi) * sc, v2 = (ai + bi) * sc, v3 = (br - ar) * sc;
        o[k] = v0; o[kBins + k] = v1; o[2 * kBins + k] = v2; o[3 * kBins + k] = v3;
        s1 += (double)v0 + (double)v1 + (double)v2 + (double)v3;
        s2 += (double)v0 * v0 + (double)v1 * v1 + (double)v2 * v2 + (double)v3 * v3;
    }
    // block reduce -> one fp64 atomic pair per workgroup, spread over slots
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if ((i & 63) == 0) { red[(i >> 6) * 2] = s1; red[(i >> 6) * 2 + 1] = s2; }
    __syncthreads();
    if (i == 0) {
        double a = red[0] + red[2] + red[4] + red[6], c = red[1] + red[3] + red[5] + red[7];
        double *dst = stats + ((size_t)b * kStatSlots + (t % kStatSlots)) * 2;
        atomicAdd(dst, a);
        atomicAdd(dst + 1, c);
    }
}

// The same transform with a workgroup WALKING a run of R consecutive frames (default; the one-frame-per-workgroup kernel above stays
// behind mi_set_transpose_tiles(1) / MI_TRANSPOSE_TILES as the A/B and the bit-identity reference): twiddles staged and the window
// read once per run instead of once per frame, the NEXT frame's samples fetched into registers (32 per thread) under the current
// frame's three passes -- every barrier inside waits for LDS traffic only --, one statistics reduction and atomic pair per run.
// Same products, same transform, same stores: the spectrogram is bit-identical; the float64 statistics are summed in another
// order (they were unordered atomics already).
// grid (ceil(T / R), B), block 256; three workgroups per CU.
__global__ __launch_bounds__(256, 3) void stft_walk_kernel(const float *__restrict__ mix, int L, int T, int R, int el, int er,
                                                           const float *__restrict__ window, const float2 *__restrict__ tw,
                                                           float *__restrict__ zt, double *__restrict__ stats) {
    MI_FFT_LDS;
    __shared__ double red[8];
    const int i = threadIdx.x, b = blockIdx.y;
    const int t_begin = blockIdx.x * R, t_end = min(T, t_begin + R);
    if (t_begin >= T) return;
    stage_twiddles(tw, twr, twi, i);
    const float *x0 = mix + (size_t)b * 2 * L, *x1 = x0 + L;
    const int L0 = L + el + er, shift = 1536 - el;           // padding rules: see stft_frames_kernel
    float wv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wv[r] = window[i + 256 * r];
    float pf[32];
    auto fetch = [&](int t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int q = t * kHop + i + 256 * r - shift;
            if (q < 0) q = -q;
            if (q >= L0) q = 2 * (L0 - 1) - q;
            const int sidx = q - el;
            const bool in = sidx >= 0 && sidx < L;
            pf[2 * r] = in ? x0[sidx] : 0.f;
            pf[2 * r + 1] = in ? x1[sidx] : 0.f;
        }
    };
    fetch(t_begin);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        int iv = i;                                          // opaque per frame: keeps the passes' LDS offsets out of the loop-invariant registers
        asm volatile("" : "+v"(iv));
        cf u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = {wv[r] * pf[2 * r], wv[r] * pf[2 * r + 1]};
        if (t + 1 < t_end) fetch(t + 1);                     // in flight under this frame's passes
        lds_barrier();                                       // twiddles staged; the previous frame's split reads are done
        stockham_pass<false, 1>(u, iv, re, im, twr, twi);
        fft_tail<false>(u, iv, re, im, twr, twi);
        float *o = zt + ((size_t)b * T + t) * 4 * kBins;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = iv + 256 * r;
            const int km = (kN - k) & (kN - 1);
            const float ar = re[lpad(k)], ai = im[lpad(k)], br = re[lpad(km)], bi = im[lpad(km)];
            const float sc = 0.5f / 64.0f;
            const float v0 = (ar + br) * sc, v1 = (ai - bi) * sc, v2 = (ai + bi) * sc, v3 = (br - ar) * sc;
            o[k] = v0; o[kBins + k] = v1; o[2 * kBins + k] = v2; o[3 * kBins + k] = v3;
            s1 += (double)v0 + (double)v1 + (double)v2 + (double)v3;
            s2 += (double)v0 * v0 + (double)v1 * v1 + (double)v2 * v2 + (double)v3 * v3;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if ((i & 63) == 0) { red[(i >> 6) * 2] = s1; red[(i >> 6) * 2 + 1] = s2; }
    __syncthreads();
    if (i == 0) {
        double a = red[0] + red[2] + red[4] + red[6], c = red[1] + red[3] + red[5] + red[7];
        double *dst = stats + ((size_t)b * kStatSlots + (blockIdx.x % kStatSlots)) * 2;
        atomicAdd(dst, a);
        atomicAdd(dst + 1, c);
    }
}

// zt[b][t][4][2048] -> x[b][4][2048][T] with optional (v - mean) * inv, 32x32 LDS tiles.
// grid (ceil(T/32), 2048/32, B*4)
__global__ __launch_bounds__(256) void cac_transpose_kernel(const float *__restrict__ zt, int T, const float2 *__restrict__ norm,
                                                            float *__restrict__ x, int Tp /* row pitch of x */) {
    __shared__ float tile[32][33];
    const int bc = blockIdx.z, b = bc >> 2, c = bc & 3;
    const int t0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    float mean = 0.f, inv = 1.f;
    if (norm) { float2 m = norm[b]; mean = m.x; inv = m.y; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r;
        if (t < T) tile[ty + 8 * r][tx] = zt[(((size_t)b * T + t) * 4 + c) * kBins + k0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + ty + 8 * r, t = t0 + tx;
        if (t < T) x[(((size_t)b * 4 + c) * kBins + k) * Tp + t] = (tile[tx][ty + 8 * r] - mean) * inv;
    }
}

// Strip form of the two transposes (default; the 32 x 32 tile kernels stay behind MI_TRANSPOSE_TILES=1 as the A/B and the
// bit-identity reference).  A workgroup owns 32 bins x up to kStripT frames of one plane.  On the conv-layout side that strip is
// ONE contiguous run of 32 rows (43 KB at T = 336), moved as 16-byte accesses: no tile of it straddles a 128-byte line.  With
// tiles, every odd row of 336 floats starts 64 bytes into a line, each of the ten inner tile boundaries of a row splits a line
// between two workgroups, and consecutive workgroups sit on different XCDs: both L2s fetched the line (PMC 1.26 x the
// algorithmic bytes for the inverse side) and 4.5 % of the lanes idled in the half-empty eleventh tile.  On the frame-major side
// a half-wave moves one aligned 128-byte run of bins, as before.  Values and arithmetic are those of the tile kernels.
constexpr int kStripT = 336;
__device__ __host__ __forceinline__ int strip_pitch(int tc) { return tc | 1; }       // odd: the bin-major column accesses spread over the banks

// KB bins x TS frames per workgroup: 32 x 336, the whole row of a segment (64 x 168 -- 256-byte runs on the frame-major side, 672-byte
// runs on the conv-layout side -- measured 153 / 548 us against 132 / 501 us).  The frame-major side is walked a few frames at a time (its runs lie 32 KB apart: batching them
// opened 64 DRAM pages per workgroup at once and measured slower); the contiguous side issues all of a thread's 16-byte accesses
// together.
// grid (2048/KB, B*4, ceil(T / TS)); dynamic LDS KB * strip_pitch(min(T, TS)) floats
template <int KB, int TS>
__global__ __launch_bounds__(256) void cac_transpose_strip_kernel(const float *__restrict__ zt, int T, const float2 *__restrict__ norm,
                                                                  float *__restrict__ x, int Tp) {
    extern __shared__ float strip[];
    constexpr int G = 256 / KB;                                             // frames per pass on the frame-major side
    const int bc = blockIdx.y, b = bc >> 2, c = bc & 3, k0 = blockIdx.x * KB;
    const int t0 = blockIdx.z * TS, tc = min(TS, T - t0), PT = strip_pitch(min(T, TS));
    const int i = threadIdx.x, tx = i % KB, ty = i / KB;
    float mean = 0.f, inv = 1.f;
    if (norm) { float2 m = norm[b]; mean = m.x; inv = m.y; }
    for (int t = ty; t < tc; t += G) strip[tx * PT + t] = zt[(((size_t)b * T + t0 + t) * 4 + c) * kBins + k0 + tx];
    __syncthreads();
    const bool vec = ((tc | Tp) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;       // t0 is a multiple of 4
    float *rows = x + ((size_t)bc * kBins + k0) * Tp + t0;
    if (vec) {
        constexpr int N4 = TS / 4, J = (KB * N4 + 255) / 256;
        const int n4 = tc >> 2;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int f = i + 256 * j, k = f / N4, t4 = f % N4;
            if (k < KB && t4 < n4) {
                const float *sr = strip + k * PT + 4 * t4;
                float4 v;
                v.x = (sr[0] - mean) * inv; v.y = (sr[1] - mean) * inv; v.z = (sr[2] - mean) * inv; v.w = (sr[3] - mean) * inv;
                reinterpret_cast<float4 *>(rows + (size_t)k * Tp)[t4] = v;
            }
        }
    } else {
        const int lane = i & 63, w = i >> 6;
#pragma unroll 1
        for (int k = w; k < KB; k += 4)
            for (int t = lane; t < tc; t += 64) rows[(size_t)k * Tp + t] = (strip[k * PT + t] - mean) * inv;
    }
}

// grid (2048/KB, B*S*4, ceil(T / TS)); dynamic LDS as above
template <int KB, int TS>
__global__ __launch_bounds__(256) void spec_transpose_strip_kernel(const float *__restrict__ y, int T, int S4, const float2 *__restrict__ denorm,
                                                                   float *__restrict__ yt, int Tp) {
    extern __shared__ float strip[];
    constexpr int G = 256 / KB;
    const int bc = blockIdx.y, b = bc / S4, sc = bc % S4, s = sc >> 2, c = sc & 3, k0 = blockIdx.x * KB;
    const int t0 = blockIdx.z * TS, tc = min(TS, T - t0), PT = strip_pitch(min(T, TS));
    const int i = threadIdx.x;
    float mean = 0.f, std = 1.f;
    if (denorm) { float2 m = denorm[b]; mean = m.x; std = m.y; }
    const bool vec = ((tc | Tp) & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    const float *rows = y + ((size_t)bc * kBins + k0) * Tp + t0;
    if (vec) {
        constexpr int N4 = TS / 4, J = (KB * N4 + 255) / 256;
        const int n4 = tc >> 2;
        float4 v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int f = i + 256 * j, k = f / N4, t4 = f % N4;
            if (k < KB && t4 < n4) v[j] = reinterpret_cast<const float4 *>(rows + (size_t)k * Tp)[t4];
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int f = i + 256 * j, k = f / N4, t4 = f % N4;
            if (k < KB && t4 < n4) {
                float *sr = strip + k * PT + 4 * t4;
                sr[0] = v[j].x * std + mean; sr[1] = v[j].y * std + mean; sr[2] = v[j].z * std + mean; sr[3] = v[j].w * std + mean;
            }
        }
    } else {
        const int lane = i & 63, w = i >> 6;
#pragma unroll 1
        for (int k = w; k < KB; k += 4)
            for (int t = lane; t < tc; t += 64) strip[k * PT + t] = rows[(size_t)k * Tp + t] * std + mean;
    }
    __syncthreads();
    const int tx = i % KB, ty = i / KB;
    for (int t = ty; t < tc; t += G) yt[((((size_t)b * (S4 >> 2) + s) * T + t0 + t) * 4 + c) * kBins + k0 + tx] = strip[tx * PT + t];
}

// bit 0: one-frame-per-workgroup STFT, bit 1: tile cac_transpose, bit 2: tile spec_transpose (mi_set_transpose_tiles(1) sets all three)
static int transpose_tiles_from_env() {
    const char *e = getenv("MI_TRANSPOSE_TILES");
    if (!e) return 0;
    const int v = atoi(e);
    return v > 1 ? (v & 7) : 7;
}
int g_transpose_tiles = transpose_tiles_from_env();

// ---------------------------------------------------------------------------------------------
// iSTFT.  (1) y[b][S*4][2048][T] (decoder output, CaC) -> frame-major yt[b][s][t][4][2048] with
//             the de-normalisation  v*std + mean  (htdemucs.py:625-626) folded in;
//         (2) one inverse FFT per (b, s, t): both channels packed, x sqrt(N)/N, x window
//             -> fr[b][s][t][2][4096];
//         (3) overlap-add gather, / window envelope, crop, + time branch * stdt + meant.
// ---------------------------------------------------------------------------------------------
// grid (ceil(T/32), 2048/32, B*S*4)
__global__ __launch_bounds__(256) void spec_transpose_kernel(const float *__restrict__ y, int T, int S4, const float2 *__restrict__ denorm,
                                                             float *__restrict__ yt, int Tp /* row pitch of y */) {
    __shared__ float tile[32][33];
    const int bc = blockIdx.z, b = bc / S4, sc = bc % S4, s = sc >> 2, c = sc & 3;
    const int t0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float mean = 0.f, std = 1.f;
    if (denorm) { float2 m = denorm[b]; mean = m.x; std = m.y; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + ty + 8 * r, t = t0 + tx;
        if (t < T) tile[ty + 8 * r][tx] = y[(((size_t)b * S4 + sc) * kBins + k) * Tp + t] * std + mean;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r;
        if (t < T) yt[((((size_t)b * (S4 >> 2) + s) * T + t) * 4 + c) * kBins + k0 + tx] = tile[tx][ty + 8 * r];
    }
}

// grid (T, B*S), block 256
__global__ __launch_bounds__(256) void istft_frames_kernel(const float *__restrict__ yt, int T, const float *__restrict__ window,
                                                           const float2 *__restrict__ tw, float *__restrict__ fr) {
    MI_FFT_LDS;
    const int i = threadIdx.x, t = blockIdx.x, bs = blockIdx.y;
    stage_twiddles(tw, twr, twi, i);
    const float *src = yt + ((size_t)bs * T + t) * 4 * kBins;
    // Z[k] = X0[k] + i X1[k];  Z[N-k] = conj(X0[k]) + i conj(X1[k]);  Nyquist bin is the zero pad of
    // htdemucs.py:444; the imaginary part of the DC bin is ignored by a C2R transform.
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = i + 256 * r;
        float ar = src[k], ai = src[kBins + k], br = src[2 * kBins + k], bi = src[3 * kBins + k];
        if (k == 0) { ai = 0.f; bi = 0.f; }
        re[lpad(k)] = ar - bi; im[lpad(k)] = ai + br;
        if (k > 0) { re[lpad(kN - k)] = ar + bi; im[lpad(kN - k)] = br - ai; }
    }
    if (i == 0) { re[lpad(kBins)] = 0.f; im[lpad(kBins)] = 0.f; }
    __syncthreads();
    cf u[16];
    load_pass_input<true>(u, i, re, im);
    __syncthreads();
    stockham_pass<true, 1>(u, i, re, im, twr, twi);
    fft_tail<true>(u, i, re, im, twr, twi);
    float *dst = fr + ((size_t)bs * T + t) * 2 * kN;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = i + 256 * r;
        const float w = window[p] * (1.0f / 64.0f);
        dst[p] = re[lpad(p)] * w;
        dst[kN + p] = im[lpad(p)] * w;
    }
}

// out[b][s][c][n] = OLA(frames)[n] / env + (xt ? xt[b][s*2+c][n] * stdt + meant : 0)
// grid (ceil(L/256), B*S*2)
__global__ __launch_bounds__(256) void istft_ola_kernel(const float *__restrict__ fr, int T, int L, const float *__restrict__ env,
                                                        const float *__restrict__ xt, const float2 *__restrict__ denorm_t, int S,
                                                        int xt_pitch, float *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const int bsc = blockIdx.y, c = bsc & 1, bs = bsc >> 1;
    // position in the un-trimmed overlap-add buffer: + 1536 (crop of htdemucs.py:449) + 2048 (centre trim)
    const int u = n + 1536 + 2048;
    const int mhi = u >> 10, r = u & 1023;
    float acc = 0.f;
#pragma unroll
    for (int j = 3; j >= 0; --j) {              // ascending frame index, like a sequential overlap-add
        const int m = mhi - j - 2;              // data frame index (frames 0,1 and T+2,T+3 are the zero pad)
        if (m >= 0 && m < T) acc += fr[(((size_t)bs * T + m) * 2 + c) * kN + r + 1024 * j];
    }
    float v = acc / env[r];
    if (xt) {
        const int b = bs / S;
        const float2 d = denorm_t[b];
        v += xt[(size_t)bsc * xt_pitch + n] * d.y + d.x;
    }
    out[(size_t)bsc * L + n] = v;
}


// Fused (2) + (3): one workgroup walks a run of consecutive frames of one (b, s) in ascending order, transforms each frame in LDS
// as istft_frames_kernel does, and keeps the overlap-add of the four hop blocks a frame touches in REGISTERS (thread i owns
// samples i + 256 q of every 1024-sample hop block, both channels: 32 accumulators).  After frame m has been added, hop block
// m + 2 holds all four of its frames (m - 3 .. m, added in ascending order: the summation order of istft_ola_kernel, so the two
// routes agree bit for bit) and goes out: / envelope, crop, + time branch.  The windowed frames (fr: S T 2 4096 floats per
// item) are never written or re-read: 2.7 GB of the 4.8 GB the two separate kernels move per batched forward; the price is three
// warm-up frames per run (the blocks they complete belong to the previous run).  The next frame's spectrum is fetched into
// registers under the current frame's passes.
// grid (runs, B*S), block 256; run g owns hop blocks [3 + g * R, 3 + (g + 1) * R).  Three workgroups per CU (117 VGPRs, 50.8 KB of LDS each).
__global__ __launch_bounds__(256, 3) void istft_fused_kernel(const float *__restrict__ yt, int T, int L, int R, const float *__restrict__ window,
                                                          const float2 *__restrict__ tw, const float *__restrict__ env,
                                                          const float *__restrict__ xt, const float2 *__restrict__ denorm_t, int S,
                                                          int xt_pitch, float *__restrict__ out) {
    MI_FFT_LDS;
    const int i = threadIdx.x, bs = blockIdx.y;
    const int hb0 = 3 + blockIdx.x * R, hb_last = (L - 1 + 3584) >> 10;        // first block of this run, last block of the signal
    const int hb1 = min(hb0 + R, hb_last + 1);
    if (hb0 > hb_last) return;
    stage_twiddles(tw, twr, twi, i);
    float ev[4];                              // this thread's envelope samples
#pragma unroll
    for (int q = 0; q < 4; ++q) ev[q] = env[i + 256 * q];
    float2 dn = make_float2(0.f, 1.f);
    if (xt) dn = denorm_t[bs / S];
    float acc[4][4][2];                       // [segment j = hop block m + 2 + j][sample i + 256 q][channel]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q][0] = acc[j][q][1] = 0.f;
    const int m_first = hb0 - 5, m_last = hb1 - 3;           // frames that touch blocks [hb0, hb1): block hb takes frames hb - 5 .. hb - 2
    // The spectrum of the NEXT frame is fetched into registers (32 per thread) while the current frame's passes run: at 117 VGPRs
    // there is room (three workgroups per CU up to 168), and the barriers between the passes wait for LDS traffic only
    float pf[32];
    auto fetch = [&](int m) {
        const float *src = yt + ((size_t)bs * T + m) * 4 * kBins;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = i + 256 * r;
            pf[4 * r] = src[k]; pf[4 * r + 1] = src[kBins + k]; pf[4 * r + 2] = src[2 * kBins + k]; pf[4 * r + 3] = src[3 * kBins + k];
        }
    };
    {
        const int m0 = max(m_first, 0);
        if (m0 <= m_last && m0 < T) fetch(m0);
    }
#pragma unroll 1
    for (int m = m_first; m <= m_last; ++m) {
        const bool live = m >= 0 && m < T;                   // workgroup-uniform: frames outside the data are the zero pad
        if (live) {
            // the thread index as the transform sees it is made opaque per frame: otherwise hipcc hoists the ~80 LDS offsets of the
            // three passes out of the frame loop as invariants and the kernel needs 236 registers (two workgroups per CU)
            int iv = i;
            asm volatile("" : "+v"(iv));
            lds_barrier();                                   // the previous frame's last LDS reads are done (and the twiddles staged)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = iv + 256 * r;
                float ar = pf[4 * r], ai = pf[4 * r + 1], br = pf[4 * r + 2], bi = pf[4 * r + 3];
                if (k == 0) { ai = 0.f; bi = 0.f; }
                re[lpad(k)] = ar - bi; im[lpad(k)] = ai + br;
                if (k > 0) { re[lpad(kN - k)] = ar + bi; im[lpad(kN - k)] = br - ai; }
            }
            if (m + 1 <= m_last && m + 1 < T) fetch(m + 1);  // in flight under this frame's three passes
            if (iv == 0) { re[lpad(kBins)] = 0.f; im[lpad(kBins)] = 0.f; }
            lds_barrier();
            cf u[16];
            load_pass_input<true>(u, iv, re, im);
            lds_barrier();
            stockham_pass<true, 1>(u, iv, re, im, twr, twi);
            fft_tail<true>(u, iv, re, im, twr, twi);
#pragma unroll
            for (int r = 0; r < 16; ++r) {                   // sample p = i + 256 r: segment r / 4, offset i + 256 (r % 4)
                const int p = iv + 256 * r;
                const float w = window[p] * (1.0f / 64.0f);          // L1 / L2 hit: 16 KB shared by every workgroup
                acc[r >> 2][r & 3][0] += re[lpad(p)] * w;
                acc[r >> 2][r & 3][1] += im[lpad(p)] * w;
            }
        }
        const int hb = m + 2;                                // complete now
        if (hb >= hb0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = hb * 1024 + i + 256 * q - 3584;
                if (n >= 0 && n < L) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        float v = acc[0][q][c] / ev[q];
                        const size_t row = (size_t)bs * 2 + c;
                        if (xt) v += xt[row * xt_pitch + n] * dn.y + dn.x;
                        out[row * L + n] = v;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[0][q][c] = acc[1][q][c]; acc[1][q][c] = acc[2][q][c]; acc[2][q][c] = acc[3][q][c]; acc[3][q][c] = 0.f;
            }
    }
}

int g_istft_fused = getenv("MI_ISTFT_SPLIT") == nullptr ? 1 : 0;

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static size_t strip_lds_bytes(int KB, int T, int TS) { return (size_t)KB * strip_pitch(std::min(T, TS)) * sizeof(float); }

int launch_stft_frames(const float *mix, int B, int L, const FftTables &tb, float *zt, double *stats, hipStream_t st) {
    const int T = ceil_div(L, kHop);
    MI_REQUIRE(L >= 1, "stft: empty input");
    // pad1d's short-input rule (hdemucs.py:29-36): zero padding before the reflection
    const int left = 1536, right = 1536 + T * kHop - L, max_pad = std::max(left, right);
    int el = 0, er = 0;
    if (L <= max_pad) { const int extra = max_pad - L + 1; er = std::min(right, extra); el = extra - er; }
    if (g_transpose_tiles & 1) {
        hipLaunchKernelGGL(stft_frames_kernel, dim3(T, B), dim3(256), 0, st, mix, L, T, el, er, tb.window, tb.twiddle, zt, stats);
    } else {
        // runs of R consecutive frames per workgroup, about two waves of workgroups over the chip's 768 slots (B = 31, T = 336: R = 7)
        const int runs = std::max(1, std::min(T, ceil_div(1536, B))), R = ceil_div(T, runs);
        hipLaunchKernelGGL(stft_walk_kernel, dim3(ceil_div(T, R), B), dim3(256), 0, st, mix, L, T, R, el, er, tb.window, tb.twiddle, zt, stats);
    }
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_cac_transpose(const float *zt, int B, int T, const float2 *norm, float *x, hipStream_t st, int x_pitch) {
    if (g_transpose_tiles & 2)
        hipLaunchKernelGGL(cac_transpose_kernel, dim3(ceil_div(T, 32), kBins / 32, B * 4), dim3(256), 0, st, zt, T, norm, x, x_pitch ? x_pitch : T);
    else
        hipLaunchKernelGGL((cac_transpose_strip_kernel<32, kStripT>), dim3(kBins / 32, B * 4, ceil_div(T, kStripT)), dim3(256), strip_lds_bytes(32, T, kStripT), st,
                           zt, T, norm, x, x_pitch ? x_pitch : T);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_istft(const float *y, int B, int S, int L, const float2 *denorm, const float *xt, const float2 *denorm_t,
                 const FftTables &tb, float *yt, float *fr, float *out, hipStream_t st, int xt_pitch, int y_pitch) {
    const int T = ceil_div(L, kHop);
    if (g_transpose_tiles & 4)
        hipLaunchKernelGGL(spec_transpose_kernel, dim3(ceil_div(T, 32), kBins / 32, B * S * 4), dim3(256), 0, st, y, T, S * 4, denorm, yt, y_pitch ? y_pitch : T);
    else
        hipLaunchKernelGGL((spec_transpose_strip_kernel<32, kStripT>), dim3(kBins / 32, B * S * 4, ceil_div(T, kStripT)), dim3(256), strip_lds_bytes(32, T, kStripT),
                           st, y, T, S * 4, denorm, yt, y_pitch ? y_pitch : T);
    MI_CHECK_LAUNCH();
    // default: the fused frame + overlap-add kernel; MI_ISTFT_SPLIT=1 / mi_set_istft_fused(0): the two separate kernels (A/B, and
    // the bit-identity test of the fused one)
    if (g_istft_fused) {
        const int blocks = ((L - 1 + 3584) >> 10) - 2;               // hop blocks 3 .. last
        // runs of ~43 blocks (three warm-up frames per run: 7 % more transforms) while that still fills the chip
        const int runs = std::max(1, std::min(ceil_div(blocks, 12), std::max(ceil_div(blocks, 43), ceil_div(1024, B * S))));
        const int R = ceil_div(blocks, runs);
        hipLaunchKernelGGL(istft_fused_kernel, dim3(ceil_div(blocks, R), B * S), dim3(256), 0, st, yt, T, L, R, tb.window, tb.twiddle, tb.envelope, xt,
                           denorm_t, S, xt_pitch ? xt_pitch : L, out);
        MI_CHECK_LAUNCH();
        return MI_OK;
    }
    hipLaunchKernelGGL(istft_frames_kernel, dim3(T, B * S), dim3(256), 0, st, yt, T, tb.window, tb.twiddle, fr);
    MI_CHECK_LAUNCH();
    hipLaunchKernelGGL(istft_ola_kernel, dim3(ceil_div(L, 256), B * S * 2), dim3(256), 0, st, fr, T, L, tb.envelope, xt, denorm_t, S, xt_pitch ? xt_pitch : L, out);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
