// Internal launcher declarations shared by the engine's translation units.
#pragma once
#include <cmath>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_conv.h"

namespace mi {

struct FftTables {
    const float *window;    // [4096] periodic Hann, float32 like th.hann_window
    const float2 *twiddle;  // [2048] fft_twiddle_table(): exp(-2 pi i n / 4096)
    const float *envelope;  // [1024] sum_j window^2[r + 1024 j]
};
// W^n = exp(-2 pi i n / 4096) for n < 2048 (W^(n + 2048) = -W^n), float32 of the float64 value: one builder for both engines and the
// handle-free entry points
static inline std::vector<float2> fft_twiddle_table() {
    std::vector<float2> tw(2048);
    for (int i = 0; i < 2048; ++i) { const double a = -2.0 * M_PI * i / 4096.0; tw[i] = make_float2((float)cos(a), (float)sin(a)); }
    return tw;
}

// fft.hip
int launch_stft_frames(const float *mix, int B, int L, const FftTables &tb, float *zt, double *stats, hipStream_t st);
int launch_cac_transpose(const float *zt, int B, int T, const float2 *norm, float *x, hipStream_t st, int x_pitch = 0 /* 0 = T */);
int launch_istft(const float *y, int B, int S, int L, const float2 *denorm, const float *xt, const float2 *denorm_t,
                 const FftTables &tb, float *yt, float *fr, float *out, hipStream_t st, int xt_pitch = 0 /* row pitch of xt, 0 = L */,
                 int y_pitch = 0 /* row pitch of y along T, 0 = T */);

extern int g_transpose_tiles;  // fft.hip: round-3 kernels around the transforms; bit 0: one STFT frame per workgroup, bit 1: tile cac_transpose, bit 2: tile spec_transpose
extern int g_istft_fused;      // fft.hip: 1 = fused inverse-transform + overlap-add kernel (default), 0 = the two separate kernels

// norms.hip
int launch_row_stats(const float *x, int rows, int64_t count, int64_t row_stride, double *stats, hipStream_t st);
int launch_finalize_stats(double *stats, int rows, double count, float eps, int mode, float2 *out_a, float2 *out_b,
                          hipStream_t st);
int launch_row_affine(const float *x, int rows, int64_t count, const float2 *norm, float *y, hipStream_t st);
int launch_row_denorm(const float *x, int rows, int64_t count, const float2 *denorm, float *y, hipStream_t st);
// img (optional, half modes): the tensor the next matrix product reads, as its 16-bit operand image [C / 8][img_n][8]
int launch_layernorm_cf(const float *x, int B, int C, int T, const float *w, const float *b, const float *pe, float *y,
                        float2 *ostat, hipStream_t st, void *img = nullptr, int64_t img_n = 0, int img_dtype = 0);
int launch_token_stats(const float *x, int B, int C, int T, float2 *ostat, hipStream_t st, void *img = nullptr, int64_t img_n = 0,
                       int img_dtype = 0);
int launch_gn_apply_tokstats(const float *x, int B, int C, int T, const float2 *gstat, const float *w, const float *b, float *y,
                             float2 *ostat, hipStream_t st, void *img = nullptr, int64_t img_n = 0, int img_dtype = 0);

// GroupNorm(1) + GELU in place plus the Gram accumulators that give the NEXT GroupNorm's statistics (norms.hip)
int gram_hp(int h);          // accumulator matrix order for h hidden channels: h + 1 rounded up to 32
int launch_gn_gelu_gram(float *x, int B, int h, int Cs, int D1, int D2 /* valid */, int pitch, int row_mode, const float2 *stats, const float *w,
                        const float *b, double *gram /* rows x slots x HP x HP, zero */, int slots, hipStream_t st);
int launch_gram_finalize(double *gram, int rows, int h, int slots, const double *wt /* HP x HP */, const double *ct /* HP */, double sum_b,
                         double sum_bsq, double cols, double count, float eps, float2 *out, hipStream_t st);
int launch_gn_gelu(float *x, int B, int C, int Cs, int D1, int D2, int row_mode, const float2 *stats, const float *w, const float *b,
                   hipStream_t st);

// dconv_row.hip: both DConv layers of one frequency-branch row fused in LDS
struct DConvRowLayer {
    const float *w0;    // [C][3][HA]   conv3 weights, hidden index fastest (HA = C/8 rounded up to 4)
    const float *b0;    // [HA]
    const float *g1w, *g1b;   // [HA]  GroupNorm(1, C/8) affine
    const float *w3;    // [2C][HA]    1x1 weights, natural row order (value rows then gate rows)
    const float *b3, *g2w, *g2b;   // [2C]
    const float *ls;    // [C]         LayerScale
};
// a DConv layer's weights plus the constants that give the second GroupNorm's statistics from the Gram matrix of the
// hidden activations (dconv_time.hip, dconv_row.hip)
struct DConvTimeLayer {
    DConvRowLayer w;             // same packing as dconv_row.hip
    const double *gram_a;        // [H (H + 1) / 2]  (W3^T W3)_ii, then 2 (W3^T W3)_ik for k > i, row-major upper triangle
    const double *gram_v;        // [H]              2 W3^T b3
    const double *gram_c;        // [H]              column sums of W3
    const double *gram_e1, *gram_e2;   // [H (H + 1) / 2 + H] the same constants in the ENTRY order of dconv_row.hip's reduction
                                 //   (i, k = i .. H): weight of the entry in sum z^2 / in sum z
    double sum_b3, sum_b3sq;
};
struct DConvRowArgs {
    DConvTimeLayer l[2];
    const float *x;     // [B][C][Fr][T]
    float *y;           // same shape (may alias x)
    int Fr, T;
};
bool dconv_row_supported(int C, int T);
int launch_dconv_row(const DConvRowArgs &a, int C, int rows, hipStream_t st);

// dconv_time.hip: fused DConv layer of the time branch (C = 48 / 96), three streaming passes
bool dconv_time_supported(int C, int Lp);
int launch_dconv_time_layer(const DConvTimeLayer &l, int C, int dil, int B, int Lv, int Lp, const float *x, float *y, float *hbuf,
                            double *stats, double *gram, float2 *st1, float2 *st2, hipStream_t st);

// gemm_conv.hip
int launch_conv(const mi_conv_desc &d, hipStream_t st);
int conv_pick_tile(int M);
// gemm_x6.hip
bool conv_x6_supported(int tile);
int launch_conv_x6(const mi_conv_desc &d, int tile, bool plain, hipStream_t st);
int launch_pack_split(const float *wt, int Kpad, int Mpad, int tile_m, void *wx, hipStream_t st);

// gemm_half.hip: bf16 / fp16 operand main loop (mi_config.dtype)
int launch_conv_half(const mi_conv_desc &d, int tile, bool plain, hipStream_t st);
int launch_pack_half(const float *wt, int Kpad, int Mpad, int dtype, void *wh, hipStream_t st);
// gemm_tap.hip: k x k stride-1 convs on operand-image inputs (tap-minor K order), half modes
int conv_tap_pairs_pad(int Cin, int ntaps);
int launch_pack_tap(const float *wt, int Mpad, int Cin, int ntaps, int dtype, void *out, hipStream_t st);
int launch_conv_tap(const mi_conv_desc &d, int tile, hipStream_t st);
int launch_f32_to_image(const float *x, int B, int C, int64_t P, int dtype, void *img, hipStream_t st);

// hkernels.hip: the Hybrid Demucs v3 (hdemucs_mmi) path's own kernels
int launch_row_affine_pitch(const float *x, int B, int C, int L, int out_pitch, const float2 *norm, float *y, hipStream_t st);
int launch_gn_apply(const float *x, int B, int Cin, int G, int in_pitch, int off, const float2 *stats, const float *w, const float *bias,
                    int glu, int gelu, const float *scale, const float *res, int res_pitch, float *y, int Cout, int out_len,
                    int out_pitch, hipStream_t st, int chan_div = 1);
int launch_unfold_frames(const float *x, int B, int C, int T, int F, int W, int S, float *fr, hipStream_t st);
int launch_restitch_frames(const float *fr, int B, int C, int T, int F, int W, int S, const float *skip, float *y, hipStream_t st);
void pack_lstm_whh(const float *whh /* (2, 4H, H) */, int H, float *packed);     // host side: the step kernel's operand order
int launch_lstm_seq(const float *gx, const float *whh /* packed */, int N, int H, int W, float *out, float *state /* 6 N H floats */, hipStream_t st);
// lstm.hip: the same recurrence as ONE persistent launch per batch of sequence tiles (hidden state exchanged between workgroups
// as tagged 8-byte granules); scratch: lstm_persist_scratch_bytes() device bytes; ctl_host: pinned host word set on a time-out
size_t lstm_persist_scratch_bytes();
size_t lstm_persist_ctl_offset();      // byte offset of the control / debug words inside the scratch block
int launch_lstm_persist(const float *gx, const float *whh /* packed */, int N, int H, int W, float *out, void *scratch, unsigned *ctl_host,
                        hipStream_t st);
int launch_lstm_small(const float *gx, const float *whh_natural /* (2, 4H, H) */, int N, int H, int W, float *out, hipStream_t st);   // H <= 64
int launch_local_attn(const float *qkc, int B, int C, int T, int ld /* row pitch of qkc, % 4 == 0 */, float *out, int ld_o, hipStream_t st);

// attention.hip
// oh != NULL (half modes): the output goes to a 16-bit operand image [512 / 8][oh_n][8] (column b * Tq + query) instead of o
int launch_attention(const float *q, const float *k, const float *v, float *o, int B, int heads, int Tq, int Tk, int64_t q_bs,
                     int64_t kv_bs, int64_t o_bs, int dtype, hipStream_t st, void *oh = nullptr, int64_t oh_n = 0);

// ola.hip
int launch_segments_gather(const float *track, int64_t track_len, int channels, const int64_t *starts_dev, int B, int valid,
                           float *seg, hipStream_t st);
int launch_ola_accumulate(float *acc, int64_t acc_len, int rows, const float *model_out, int valid, const int64_t *offs_dev,
                          const int32_t *lens_dev, const int32_t *trim_dev, int B, int64_t span_lo, int64_t span_hi,
                          const float *weight, int weight_len, hipStream_t st);
int launch_ola_finish(float *acc, int64_t acc_len, int rows, int64_t acc_off0, const int64_t *offs_dev, const int32_t *lens_dev,
                      int n_segments, int max_len, const float *weight, hipStream_t st);

// resample.hip
int launch_resample_frac(const float *x, int rows, int64_t L, const float *table, int old_sr, int new_sr, int width, float *y,
                         int64_t Lout, hipStream_t st);

const void *conv_zero_page();      // gemm_conv.hip: 256 bytes of device zeros
// attention_heads.hip: half modes, per-head token-major 16-bit operands (MI_FLAG_HEADS), LDS-DMA ring
int launch_attention_heads(const void *q, const void *k, const void *v, const void *zero_page, int B, int heads, int Tq, int Tk, int Tq_pitch,
                           int Tk_pitch, int dtype, void *oh, int64_t oh_n, float *o, int64_t o_bs, hipStream_t st);

// post.hip: Separator normalisation, clip prevention, two-stems sums
int post_stats_scratch_bytes();
int launch_mono_stats(const float *wav, int channels, int64_t length, double *scratch, float *stats, hipStream_t st);
int launch_track_affine(float *x, int64_t n, const float *stats, int mode, hipStream_t st);
int launch_prevent_clip(const float *x, int64_t n, int mode, unsigned *peak, float *y, hipStream_t st);
int launch_two_stems(const float *const *stems, int S, int sel, const float *origin, int mode, int64_t n, float *y, hipStream_t st);

}  // namespace mi
