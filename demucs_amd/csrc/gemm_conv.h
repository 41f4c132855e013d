// Implicit-GEMM convolution / linear layer on the fp32 MFMA path (v_mfma_f32_32x32x2_f32).
//
//   Y[m][n] = epilogue( sum_k Wt[k][m] * B[k][n] ),   B[k][n] = prologue( X[gather(k, n)] )
//
// n enumerates output positions (b, o1, o2) of a channel-first activation tensor, k enumerates
// (input channel, tap).  The gather is table driven (one int4 per k), so ONE kernel covers
// Conv1d/Conv2d with stride / dilation / padding, 1x1 convs and nn.Linear on channel-first
// tokens, and ConvTranspose(k=8, s=4) rewritten as a 4-phase GEMM with a scatter epilogue.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum mi_epilogue {
    MI_EPI_LINEAR = 0,      /* y = [res +] [scale *] act(acc + bias)                                   */
    MI_EPI_GLU = 1,         /* rows interleaved (a_c, g_c): y[c] = (a+ba) * sigmoid(g+bg) [+ emb[c][o1]] */
    MI_EPI_BIAS_STATS = 2,  /* y = acc + bias, and per-row sum / sum-of-squares -> stats (fp64 atomics) */
    MI_EPI_STATS_ONLY = 3,  /* statistics of (acc + bias) only, nothing stored                          */
    MI_EPI_GN_GLU = 4,      /* y[c] = res + scale[c] * GLU(GroupNorm(acc + bias))  (DConv tail)         */
    MI_EPI_CONVTR = 5       /* rows (co, phase r): y[co][s*o + r - pad] = [res +] act(acc + bias), s = tr_stride (default 4), pad = tr_pad */
};

#define MI_FLAG_GELU 1
#define MI_FLAG_SCALE 2
#define MI_FLAG_RES 4
#define MI_FLAG_EMB 8
#define MI_FLAG_TR_FREQ 16 /* CONVTR scatters along o1 (frequency axis) instead of o2 (time axis) */
/* LINEAR on LayerNorm-ed tokens without materialising them: with W' = W diag(ln_w), the caller packs W' as
 * the weights, c1[m] = sum_k W'[m][k] in `scale`, c2[m] = (W ln_b)[m] + bias[m] in `bias`, and passes the
 * per-column (mean, rstd) of the RAW input in `pro_stats` (float2 per output column):
 *     y = act( rstd[n] * (acc - mean[n] * c1[m]) + c2[m] )                                      */
#define MI_FLAG_LN 32
#define MI_FLAG_IMG 64     /* LINEAR or GLU (M % 32 == 0), half modes: the result goes to `yh` as a 16-bit operand image instead of `y` */
/* LINEAR, half modes, attention projections (M = 512 n rows, O1 = 1): the result goes to `yh` ONLY, as 16-bit per-head
 * token-major tensors  yh[row / 512][b][head = (row / 64) % 8][token (pitch yh_n)][row % 64]  -- the operands of
 * attention_heads.hip (Q, K, V are consumed by nothing else: no float32 copy is written) */
#define MI_FLAG_HEADS 128
/* LINEAR: also accumulate sum / sum of squares of the stored result per item b into `stats` (float64 atomics, kStatSlots slots:
 * the GroupNorm that follows a transformer layer needs no pass of its own over the tensor); O2 >= 32 */
#define MI_FLAG_STATS 256
/* GLU epilogue, half modes: besides y, the result goes to `yh` as the PHASE-SPLIT operand image of the next encoder conv (k = 8,
 * s = 4, pad 2 along o1 with MI_FLAG_TR_FREQ, else along o2): index i of that axis lands in plane rho = i % 4 at slot
 * q = i / 4 + (rho >= 2), image [(channel octet * 4 + rho)][b * yh_pq + q (* O2 + o2)][8].  Output o of the strided conv then reads
 * slots o and o + 1 of every plane -- taps 2, 6 / 3, 7 / 0, 4 / 1, 5 for rho = 0 .. 3 -- i.e. it is a stride-1 two-tap conv over
 * 4 C channels that gemm_tap.hip runs by LDS-DMA.  Slots never written (q = 0 of planes 2, 3; past the end) must be zero. */
#define MI_FLAG_IMG4 512
/* MI_FLAG_IMG on a CONVTR epilogue (half modes): the scattered result goes to `yh` as the operand image
 * [Cout / 8][yh_n positions][8] of the NEXT layer's k x k conv (position = b * y_cstride + scattered index); y is not written */

typedef struct mi_ktab_entry {
    int32_t off; /* element offset added to the column base: ci*chan_stride + d1*D2 + d2 */
    int32_t d1;  /* i1 = o1*S1 + d1 must lie in [0, D1)                                   */
    int32_t d2;  /* i2 = o2*S2 + d2 must lie in [0, D2)                                   */
    int32_t ci;  /* input channel (for the per-channel prologue affine)                    */
} mi_ktab_entry;

typedef struct mi_conv_desc {
    /* weights, packed [Kpad][Mpad] (M contiguous), zero padded */
    const float *wt;
    int32_t M, Mpad, K, Kpad;
    const mi_ktab_entry *ktab; /* [Kpad] device */
    /* input activation X[b][Cin][D1][D2] */
    const float *x;
    int64_t x_bstride;
    int32_t B, D1, D2, O1, O2, S1, S2;
    /* prologue: must be 0 (the GroupNorm+GELU of the DConv hidden tensor runs as its own pass) */
    int32_t pro;
    const float *pro_stats; /* float2 [rows] (mean, rstd) */
    const float *pro_w, *pro_b;
    int32_t row_mode; /* statistics row of a column: 0 -> b, 1 -> b*O1 + o1 */
    /* epilogue */
    int32_t epi, flags;
    const float *bias;  /* [Mpad] packed row order */
    const float *scale; /* LINEAR: [Mpad]; GN_GLU: [Cout] */
    const float *res;   /* same layout as y */
    const float *emb;   /* GLU + MI_FLAG_EMB: [Cout][O1] */
    float *y;
    int64_t y_bstride, y_cstride;
    double *stats;          /* [rows][32][2] fp64 partial sums (zeroed by the caller)   */
    const float *gn_stats;  /* float2 [rows] (mean, rstd) for MI_EPI_GN_GLU             */
    const float *gn_w, *gn_b; /* [Mpad] packed row order                                 */
    int32_t out_len;        /* CONVTR: valid output length along the scattered axis      */
    int32_t tile_m;         /* 0 = choose automatically; else 32 / 64 / 96 / 128         */
    int32_t plain;          /* 1: the gather is the identity over input channels (1x1 conv / linear with channel
                               stride O1*O2): enables the table-free float4 loader when shapes allow             */
    int32_t half;           /* 0, or MI_DTYPE_BF16 / MI_DTYPE_F16 (include/demucs_amd.h): run the main loop of gemm_half.hip on `wh` */
    int32_t o2_valid;       /* 0, or the number of REAL positions along o2 when O2 is a padded row pitch (columns with
                               o2 >= o2_valid are computed but never stored nor counted): lets rows whose length is
                               not a multiple of 4 keep 16-byte aligned starts and use the float4 loader            */
    int32_t ktab_len;       /* entries in ktab (0 = Kpad): the half-precision main loop steps K by 32 and needs Kpad rounded up to 32 */
    float *sink;            /* >= 256 floats that out-of-range epilogue stores are diverted to; NULL = library-owned */
    const void *wx;         /* NULL, or the weights as the split-bf16 tile image of mi_conv_pack_split for THIS tile_m:
                               selects the 6-product bf16 MFMA main loop (gemm_x6.hip) for tile_m 64 / 96 / 128     */
    int32_t tr_stride;      /* CONVTR: 0 (= 4, crop 2: ConvTranspose k = 8, s = 4 with the reference's crop folded in), 4 or 2 */
    int32_t tr_pad;         /* CONVTR with tr_stride != 0: samples cropped from the front (0 = the un-cropped transposed conv)   */
    int32_t x_ld;           /* 0, or the row pitch of X along D2 in floats when it differs from D2 (padded frequency-branch rows of
                               the hdemucs engine: D2 stays the VALID length the gather bounds test against)                   */
    int32_t x_ld_pad;
    const void *wh;         /* `half` != 0: the weights as Wh[ceil(Kpad/32)*4][Mpad][8] bf16 / fp16 (mi_conv_pack_half)           */
    /* 16-bit operand images of GEMM-only activations (half modes, plain LINEAR layers): Img[rows / 8][n_img][8] with the
       column index n = b * O1 * O2 + p of the tensor -- the order a K step's B tile has in LDS, so a consumer moves it
       global -> LDS by DMA without conversion */
    const void *xh;         /* NULL, or the INPUT as such an image (K % 8 == 0): x is then ignored                              */
    int64_t xh_n;           /* its column count                                                                               */
    void *yh;               /* MI_FLAG_IMG: the OUTPUT image (M % 8 == 0); y is not written                                      */
    int64_t yh_n;
    /* k x k stride-1 conv on an operand-image input (gemm_tap.hip): the weights with k ordered (channel octet, tap, channel % 8)
       (mi_conv_pack_tap) and the kernel geometry; xh must be set, x / ktab are ignored */
    const void *wtap;
    int32_t ntaps, tap_k2, tap_pad1, tap_pad2; /* K1 * K2 taps, K2 columns per kernel row, padding along d1 / d2                  */
    int32_t tap_dil1, tap_dil2;                /* tap step along d1 / d2: 0 = 1; -1 for the two taps of a transposed conv (input q - j) */
    int64_t yh_pq;                             /* MI_FLAG_IMG4: positions per item and plane of the phase-split output image        */
    int32_t dma_rows;       /* float32 layers: 1 = the caller vouches that EVERY entry of `ktab` has d2 == 0 (taps move along rows only:
                               strided frequency-axis convs, frequency-axis transposed convs): admits the LDS-DMA main loop of
                               conv_gemm_dmarow_kernel (bit-identical results); 0 = table-driven gather                              */
    int32_t dma_rows_pad;
} mi_conv_desc;

#ifdef __cplusplus
}
#endif
