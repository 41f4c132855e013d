// Device-side pieces shared by the implicit-GEMM main loops (gemm_conv.hip: fp32 MFMA; gemm_x6.hip: split-bf16
// MFMA): tile constants, output-column decomposition, the table-driven gather and the fused epilogues.
#pragma once
#include "common.h"
#include "gemm_conv.h"
#include "kernels.h"

namespace mi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int BN = 128;

struct ColInfo {   // decomposition of one output column n = (b, o1, o2)
    int b, o1, o2, p;
    bool valid;
};

__device__ __forceinline__ ColInfo decompose(int n, int N, int P, int O2, int o2v) {
    ColInfo c;
    c.valid = n < N;
    const int nn = c.valid ? n : 0;
    c.b = nn / P;
    c.p = nn - c.b * P;
    c.o1 = c.p / O2;
    c.o2 = c.p - c.o1 * O2;
    c.valid = c.valid && c.o2 < o2v;       // o2v < O2: O2 is a padded row pitch
    return c;
}

// the 16 rows a lane owns in one 32x32 accumulator tile are 4 groups of 4 consecutive rows: fetch a per-row
// vector (bias, scale, GroupNorm affine) for them with 4 float4 loads instead of 16 dependent dword loads
__device__ __forceinline__ void load_rows16(const float *p, int mbase, float (&out)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4 *>(p + mbase + 8 * g);
        out[4 * g] = v.x; out[4 * g + 1] = v.y; out[4 * g + 2] = v.z; out[4 * g + 3] = v.w;
    }
}

// B-operand gather for one element: branch-free (invalid taps read x[0] and are zeroed by a select)
__device__ __forceinline__ float gather_b(const mi_conv_desc &d, const mi_ktab_entry e, const float *xcol, int i1b, int i2b,
                                          bool colvalid, bool &ok) {
    const int i1 = i1b + e.d1, i2 = i2b + e.d2;
    ok = colvalid && (unsigned)i1 < (unsigned)d.D1 && (unsigned)i2 < (unsigned)d.D2;
    const float *p = ok ? xcol + e.off : d.x;
    return *p;
}

// Transposed-conv scatter of one 32x32 accumulator tile (stride 2^LG, kernel 2 * stride): row m = (co, phase), column q ->
// output index o = stride * q + phase - pad of channel co, cropped to [0, out_len).  Branch-free index arithmetic: the column's
// per-phase offsets `po` and the tile's per-channel offsets are 32-bit and carry "invalid" as a large negative number, so one add
// gives the offset and its sign the validity; the 16 skip loads are issued together ahead of the arithmetic.  (The first version
// -- 64-bit index products and short-circuit conditions per element -- cost 64 VALU instructions per value, three times the
// main loop of a K = 768 tile.)
struct TrCol {          // column-constant part, computed once per output column
    int po[4];          // o * ostep per phase, or kTrBad
    size_t colbase;     // b * y_bstride (+ o2 on the frequency axis)
    size_t imgbase;     // (b * y_cstride (+ o2)) * 8: position part of the operand-image index, in 16-bit units
};
constexpr int kTrBad = -(1 << 30);

template <int LG>
__device__ __forceinline__ TrCol convtr_column(const mi_conv_desc &d, const ColInfo &c) {
    const bool trf = d.flags & MI_FLAG_TR_FREQ;
    const int tpad = d.tr_stride ? d.tr_pad : 2, ostep = trf ? d.O2 : 1;
    const int oq = ((trf ? c.o1 : c.o2) << LG) - tpad;
    TrCol t;
#pragma unroll
    for (int ph = 0; ph < (1 << LG); ++ph) {
        const int o = oq + ph;
        t.po[ph] = (c.valid && (unsigned)o < (unsigned)d.out_len) ? o * ostep : kTrBad;
    }
    const int o2 = trf ? c.o2 : 0;
    t.colbase = c.valid ? (size_t)c.b * d.y_bstride + o2 : 0;
    t.imgbase = ((size_t)c.b * d.y_cstride + o2) * 8;
    return t;
}

// F: the GELU / RES / IMG bits as compile-time constants (with runtime flags hipcc turned the per-element conditions into
// control flow again); the host admits the three combinations the models use (launch_conv)
template <int LG, int F>
__device__ __forceinline__ void convtr_tile(const mi_conv_desc &d, const f32x16 &acc, const float (&biasr)[16], const TrCol &t, int mbase,
                                            float *sink) {
    const int cs = (int)d.y_cstride;
    constexpr bool gelu = F & MI_FLAG_GELU, resf = F & MI_FLAG_RES, imgf = F & MI_FLAG_IMG;
    // rows of the tile: m = mbase + j, j = (r & 3) + 8 * (r >> 2), mbase % 4 == 0 -> co = co0 + (j >> LG), phase = j & (stride - 1)
    const int co0 = mbase >> LG, cout = d.M >> LG;
    float *const ycol = d.y + t.colbase;
    const float *const rcol = d.res + t.colbase;
    unsigned short *const yimg = reinterpret_cast<unsigned short *>(d.yh) + t.imgbase;
    // eight rows at a time: with all 16 offsets (and their clamped 64-bit forms for the residual loads) live at once the 96-row
    // tile spilled 104 bytes per lane at four workgroups per CU
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int off[8];
        float resv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * h + q, j = (r & 3) + 8 * (r >> 2), co = co0 + (j >> LG);
            const int rowoff = co < cout ? co * cs : kTrBad;          // one multiply per distinct channel after CSE
            off[q] = rowoff + t.po[j & ((1 << LG) - 1)];
        }
        if (resf) {
#pragma unroll
            for (int q = 0; q < 8; ++q) resv[q] = rcol[max(off[q], 0)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * h + q;
            float v = acc[r] + biasr[r];
            if (gelu) v = gelu_exact(v);
            if (resf) v += resv[q];
            if (imgf) {
                // the only reader is the next layer's k x k conv (gemm_tap.hip): 16-bit, [co / 8][position][8]
                const int j = (r & 3) + 8 * (r >> 2), co = co0 + (j >> LG);
                const size_t chan = (size_t)(co >> 3) * d.yh_n * 8 + (co & 7);
                const int pos8 = (off[q] - co * cs) * 8;
                if (off[q] >= 0) yimg[chan + pos8] = (unsigned short)(pack_half2(d.half, v, 0.f) & 0xffffu);
            } else {
                *(off[q] >= 0 ? ycol + off[q] : sink) = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// PLAIN = 1x1 / linear layer with K % 16 == 0 and (O1*O2) % 4 == 0: no gather table, float4 activation loads.
// Epilogue shared by the register-staged and the LDS-DMA main loops.
// acc[a][b][r] is C[m][n] with n = n0 + (wn*TN + b)*32 + li, m = m0 + (wm*TM + a)*32 + (r & 3) + 8 * (r >> 2) + 4 * lh
template <int TM, int TN, int EPI, int LFLAGS>
__device__ __forceinline__ void conv_epilogue(const mi_conv_desc &d, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int N,
                                              int P, int o2v) {
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int slot = blockIdx.x % kStatSlots;
    float *const sink = d.sink + tid;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + (wn * TN + b) * 32 + li;
        const ColInfo c = decompose(n, N, P, d.O2, o2v);
        const int row = d.row_mode ? c.b * d.O1 + c.o1 : c.b;
        float s1 = 0.f, s2 = 0.f;
        float2 lnstat = make_float2(0.f, 1.f);
        if (EPI == MI_EPI_LINEAR && (LFLAGS & MI_FLAG_LN))
            lnstat = reinterpret_cast<const float2 *>(d.pro_stats)[c.valid ? n : 0];
        float gmean = 0.f, grstd = 0.f;
        if (EPI == MI_EPI_GN_GLU) {
            const float2 st = reinterpret_cast<const float2 *>(d.gn_stats)[row];
            gmean = st.x; grstd = st.y;
        }
        const int tr_lg = d.tr_stride == 2 ? 1 : 2;
        TrCol trc;
        if (EPI == MI_EPI_CONVTR) trc = tr_lg == 2 ? convtr_column<2>(d, c) : convtr_column<1>(d, c);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            // keep the epilogue of one 32x32 accumulator tile together: without the fence hipcc copies all
            // accumulators out of the AGPR file first and the VGPR allocation (not the main loop) caps occupancy
            __builtin_amdgcn_sched_barrier(0);
            int mbase = m0 + (wm * TM + a) * 32 + 4 * lh;
            // opaque per accumulator tile: otherwise the row offsets m * cs, the bounds tests and the per-row vector addresses of ALL
            // TM row blocks are computed once, shared between the TN column blocks and kept live across the whole epilogue --
            // more address registers than the budget allows next to a tile's working set, so they went to scratch
            asm volatile("" : "+v"(mbase));
            const f32x16 &tacc = acc[a][b];
            float biasr[16], auxr[16], aux2r[16];        // this tile's per-row vectors (float4 loads, L1/L2 hits)
            load_rows16(d.bias, mbase, biasr);
            if (EPI == MI_EPI_LINEAR && (LFLAGS & (MI_FLAG_SCALE | MI_FLAG_LN))) load_rows16(d.scale, mbase, auxr);
            if (EPI == MI_EPI_GN_GLU) { load_rows16(d.gn_w, mbase, auxr); load_rows16(d.gn_b, mbase, aux2r); }
            if (EPI == MI_EPI_LINEAR) {
                // 32-bit row offsets from one 64-bit column base; for residual epilogues all 16 residual loads of
                // the tile are issued first, so they are in flight together instead of one load -> store round
                // trip per value
                const size_t colbase = c.valid ? (size_t)c.b * d.y_bstride + c.p : 0;
                float *const ycol = d.y + colbase;
                const int cs = (int)d.y_cstride;
                float resv[16], vq[4];
                if (LFLAGS & MI_FLAG_RES) {
                    const float *const rcol = d.res + colbase;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mbase + (r & 3) + 8 * (r >> 2);
                        resv[r] = rcol[(c.valid && m < d.M) ? m * cs : 0];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    float v;
                    if (LFLAGS & MI_FLAG_LN) v = lnstat.y * (tacc[r] - lnstat.x * auxr[r]) + biasr[r];
                    else v = tacc[r] + biasr[r];
                    if (LFLAGS & MI_FLAG_GELU) v = gelu_exact(v);
                    if (LFLAGS & MI_FLAG_SCALE) v *= auxr[r];
                    if (LFLAGS & MI_FLAG_RES) v += resv[r];
                    if (LFLAGS & MI_FLAG_STATS) { const bool ok = c.valid && m < d.M; s1 += ok ? v : 0.f; s2 += ok ? v * v : 0.f; }
                    if (LFLAGS & MI_FLAG_HEADS) {
                        // four consecutive channels of one head and token: one 8-byte store into that token's 128-byte row
                        vq[r & 3] = v;
                        if ((r & 3) == 3) {
                            const int mo = mbase + 8 * (r >> 2);
                            const size_t plane = ((size_t)(mo >> 9) * d.B + c.b) * 8 + ((mo >> 6) & 7);
                            uint2 *dst = reinterpret_cast<uint2 *>(d.yh) + ((plane * d.yh_n + c.p) * 64 + (mo & 63)) / 4;
                            if (c.valid && mo < d.M) *dst = make_uint2(pack_half2(d.half, vq[0], vq[1]), pack_half2(d.half, vq[2], vq[3]));
                        }
                        continue;
                    }
                    if (LFLAGS & MI_FLAG_IMG) {
                        // four consecutive rows of a k-octet of the NEXT layer's operand: one 8-byte store into its image
                        vq[r & 3] = v;
                        if ((r & 3) == 3) {
                            const int mo = mbase + 8 * (r >> 2);               // = 8 * octet + 4 * lh
                            uint2 *dst = reinterpret_cast<uint2 *>(d.yh) + ((size_t)(mo >> 3) * d.yh_n + n) * 2 + lh;
                            if (c.valid && mo < d.M) *dst = make_uint2(pack_half2(d.half, vq[0], vq[1]), pack_half2(d.half, vq[2], vq[3]));
                        }
                        continue;
                    }
                    // branch-free: out-of-range rows / columns are stored to a per-lane sink word
                    *((c.valid && m < d.M) ? ycol + m * cs : sink) = v;
                }
            } else if (EPI == MI_EPI_GLU || EPI == MI_EPI_GN_GLU) {
                // as above: one 64-bit column base, 32-bit row offsets, and the tile's eight residual / scale / embedding loads
                // issued together ahead of the arithmetic (a load -> store round trip per value is latency, not bandwidth)
                const size_t colbase = c.valid ? (size_t)c.b * d.y_bstride + c.p : 0;
                float *const ycol = d.y + colbase;
                const int cs = (int)d.y_cstride;
                const bool emb = EPI == MI_EPI_GLU && (d.flags & MI_FLAG_EMB);
                // MI_FLAG_IMG on a GLU layer (half modes): the result feeds nothing but the next matrix product (hdemucs decoders:
                // rewrite + GLU -> transposed conv, no DConv between): it goes ONLY to `yh` as the plain operand image
                // [channel octet][yh_n columns][8]; y is not written
                const bool imgp = EPI == MI_EPI_GLU && (d.flags & MI_FLAG_IMG);
                const bool img4 = (EPI == MI_EPI_GLU && (d.flags & MI_FLAG_IMG4)) || imgp;
                // phase-split operand image of the next strided conv (MI_FLAG_IMG4): plane and slot of this column
                unsigned *img4p = nullptr;
                size_t oct_step = 0;                               // dwords between the images of consecutive channel octets
                if (imgp) {
                    img4p = reinterpret_cast<unsigned *>(d.yh) + (size_t)(c.valid ? n : 0) * 4;
                    oct_step = (size_t)4 * d.yh_n;
                } else if (img4) {
                    const bool trf = d.flags & MI_FLAG_TR_FREQ;
                    const int idx = trf ? c.o1 : c.o2, rho = idx & 3, q = (idx >> 2) + (rho >> 1);
                    const size_t pos = (size_t)c.b * d.yh_pq + (trf ? (size_t)q * d.O2 + c.o2 : (size_t)q);
                    img4p = reinterpret_cast<unsigned *>(d.yh) + ((size_t)rho * d.yh_n + pos) * 4;     // 4 dwords per position
                    oct_step = (size_t)16 * d.yh_n;
                }
                float resv[8], scv[8], vprev = 0.f;
                unsigned pk4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2), ch = m >> 1;
                    const bool ok = c.valid && m < d.M;
                    if (EPI == MI_EPI_GN_GLU) { resv[r >> 1] = d.res[colbase + (ok ? ch * cs : 0)]; scv[r >> 1] = d.scale[ok ? ch : 0]; }
                    else if (emb) resv[r >> 1] = d.emb[ok ? ch * d.O1 + c.o1 : 0];
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);   // even row: value, m+1: gate
                    float va = tacc[r] + biasr[r], vg = tacc[r + 1] + biasr[r + 1];
                    if (EPI == MI_EPI_GN_GLU) {
                        va = (va - gmean) * grstd * auxr[r] + aux2r[r];
                        vg = (vg - gmean) * grstd * auxr[r + 1] + aux2r[r + 1];
                    }
                    float v = va * sigmoid_f(vg);
                    if (EPI == MI_EPI_GN_GLU) v = resv[r >> 1] + scv[r >> 1] * v;
                    else if (emb) v += resv[r >> 1];
                    *((c.valid && m < d.M && !imgp) ? ycol + (m >> 1) * cs : sink) = v;
                    if (img4) {                                // channels ch, ch + 1 (ch even) are rows r = 4 j, 4 j + 2 of this lane
                        if (r & 2) pk4[r >> 2] = pack_half2(d.half, vprev, v);
                        else vprev = v;
                    }
                }
                if (img4) {
                    // the tile's 16 channels = two octets; a lane holds pairs (0,1) (4,5) (8,9) (12,13) + 2 lh.  Two v_permlane32_swap
                    // give the lower half-wave octet 0 and the upper one octet 1 complete: one 16-byte store per lane
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    const u2 s02 = __builtin_amdgcn_permlane32_swap(pk4[0], pk4[2], false, false);
                    const u2 s13 = __builtin_amdgcn_permlane32_swap(pk4[1], pk4[3], false, false);
                    const int oct = ((mbase - 4 * lh) >> 4) + lh;
                    if (c.valid && mbase - 4 * lh + 31 < d.M)
                        *reinterpret_cast<uint4 *>(img4p + (size_t)oct * oct_step) = make_uint4(s02[0], s02[1], s13[0], s13[1]);
                }
            } else if (EPI == MI_EPI_BIAS_STATS || EPI == MI_EPI_STATS_ONLY) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    const float v = tacc[r] + biasr[r];
                    const bool ok = c.valid && m < d.M;
                    if (EPI == MI_EPI_BIAS_STATS)
                        *(ok ? d.y + ((size_t)c.b * d.y_bstride + (size_t)m * d.y_cstride + c.p) : sink) = v;
                    s1 += ok ? v : 0.f; s2 += ok ? v * v : 0.f;
                }
            } else if (EPI == MI_EPI_CONVTR) {
                if (tr_lg == 1) convtr_tile<1, 0>(d, tacc, biasr, trc, mbase, sink);
                else if (!(d.flags & MI_FLAG_GELU)) convtr_tile<2, 0>(d, tacc, biasr, trc, mbase, sink);
                else if (!(d.flags & MI_FLAG_IMG)) convtr_tile<2, MI_FLAG_GELU | MI_FLAG_RES>(d, tacc, biasr, trc, mbase, sink);
                else convtr_tile<2, MI_FLAG_GELU | MI_FLAG_RES | MI_FLAG_IMG>(d, tacc, biasr, trc, mbase, sink);
            }
        }
        if (EPI == MI_EPI_BIAS_STATS || EPI == MI_EPI_STATS_ONLY || (EPI == MI_EPI_LINEAR && (LFLAGS & MI_FLAG_STATS))) {
            // a wave's 32 columns span at most two statistics rows (O2 >= 32): reduce both groups
            double t1 = (double)s1, t2 = (double)s2;
            t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);          // the two lane halves share a column
            const int rid = c.valid ? row : -1;
            const int row0 = __shfl(rid, 0);
            int rowB = rid;
            double a1 = (rid == row0 && rid >= 0) ? t1 : 0.0, a2 = (rid == row0 && rid >= 0) ? t2 : 0.0;
            double b1 = (rid != row0 && rid >= 0) ? t1 : 0.0, b2 = (rid != row0 && rid >= 0) ? t2 : 0.0;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off);
                b1 += __shfl_xor(b1, off); b2 += __shfl_xor(b2, off);
                rowB = max(rowB, __shfl_xor(rowB, off));
            }
            if (lane == 0 && row0 >= 0) {
                double *dst = d.stats + ((size_t)row0 * kStatSlots + slot) * 2;
                atomicAdd(dst, a1); atomicAdd(dst + 1, a2);
                if (rowB != row0) {
                    dst = d.stats + ((size_t)rowB * kStatSlots + slot) * 2;
                    atomicAdd(dst, b1); atomicAdd(dst + 1, b2);
                }
            }
        }
    }
}

// Tile order for the 8 per-XCD L2s (workgroups id, id + 8, ... share one): XCD x = id & 7 belongs to M group x % Gm
// and N group x / Gm and only ever touches the weights of ITS M tiles (MT / Gm of them: sized by the host to stay L2
// resident), while consecutive workgroups of an XCD walk those M tiles for one N tile (shared activation tile).
// With Gm = 1 every XCD streams the whole weight matrix once per N tile; when that exceeds the 4 MiB L2, half of
// all L2 requests miss (TCC_HIT / TCC_MISS, profiles/).  Returns false for the grid's padding workgroups.
__device__ __forceinline__ bool tile_of_block(int MT, int Gm, int N, int &mt, int &nt) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int MTx = MT / Gm, mi = j % MTx, ni = j / MTx;
    mt = x % Gm + Gm * mi;
    nt = x / Gm + (8 / Gm) * ni;
    return nt * BN < N;
}
// host side: fewest M groups (1, 2, 4, 8; dividing MT) that bring one group's weights under 1.5 MiB, and the grid
static inline int pick_m_groups(int MT, size_t weight_bytes) {
    int Gm = 1;
    while (Gm < 8 && MT % (2 * Gm) == 0 && weight_bytes / Gm > ((size_t)3 << 19)) Gm *= 2;
    return Gm;
}
static inline unsigned grouped_grid(int MT, int NT, int Gm) { return 8u * (MT / Gm) * ((NT + 8 / Gm - 1) / (8 / Gm)); }

}  // namespace mi
