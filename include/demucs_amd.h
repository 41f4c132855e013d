/* demucs_amd.h — C ABI of the MI355X (gfx950) HTDemucs segment-inference engine.
 *
 * The upstream project (DrorT/demucs) is pure Python and has no FFI; its boundary for this
 * path is a Python call + a weight schema.  Each entry point below names the reference
 * interface it stands in for (paths relative to the reference checkout).  A maintainer binds
 * them with ctypes (demucs_amd/_lib.py is that binding).
 *
 * Conventions
 *   - every function returns 0 on success or a negative MI_E* code; nothing throws across the
 *     ABI; `mi_last_error()` returns a thread-local, NUL-terminated description of the last
 *     failure on the calling thread.
 *   - `*_dev` pointers are device (HBM) addresses owned by the caller (e.g. torch tensors'
 *     data_ptr()), contiguous float32 unless stated; `stream` is a hipStream_t (NULL = the
 *     default stream).  All work is enqueued asynchronously on `stream`; no call synchronises
 *     the device except mi_model_create / mi_model_destroy.
 *   - a handle is single-stream: do not use one handle from two streams/threads at once.
 *   - PROCESS-WIDE STATE (one process per GPU is the deployment).  Handles are independent EXCEPT for: (1) the activation
 *     workspace of the htdemucs engine, which every handle created on the same device with the same (n_sources,
 *     segment_length, max_batch, float32 / half) shares -- a bag of four fine-tuned models holds its ~0.57 GB per batched
 *     segment once: such handles must run one after the other on one stream (what apply_model's bag loop does), never
 *     concurrently; hdemucs handles own their workspace; (2) the schedule switches mi_set_two_streams / mi_set_istft_fused / mi_set_transpose_tiles
 *     and the MI_* environment variables (read once); (3) two small device blocks every launch may use: a 256-float sink for
 *     masked stores and 256 bytes of zeros (the source of out-of-frame LDS-DMA transfers).  A forward also uses a side stream
 *     owned by its handle; it is joined on the caller's stream before the call returns.
 */
#ifndef DEMUCS_AMD_H
#define DEMUCS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument / unsupported configuration            */
#define MI_EHIP (-2)     /* a HIP runtime call failed (see mi_last_error)        */
#define MI_ENOMEM (-3)   /* device or host allocation failed                     */
#define MI_EWEIGHT (-4)  /* missing / mis-shaped tensor in the weight table      */

/* One tensor of a reference `state_dict()` (demucs/states.py:83-107 yields these names;
 * schema in SURVEY.md App. B).  `data` is a HOST pointer to float32, row-major, `numel`
 * elements; the library copies what it needs before mi_model_create returns. */
typedef struct mi_tensor_desc {
    const char *name;
    const float *data;
    int64_t numel;
} mi_tensor_desc;

/* Architecture = the released htdemucs family (demucs/htdemucs.py:56-133 with
 * channels=48, depth=4, nfft=4096, dconv_mode=3, bottom_channels=512, t_layers=5, t_heads=8). */
typedef struct mi_config {
    int32_t n_sources;       /* len(model.sources): 4 (htdemucs, htdemucs_ft) or 6 (htdemucs_6s) */
    int32_t segment_length;  /* int(model.segment * model.samplerate) = 343980                   */
    int32_t max_batch;       /* segments per forward the workspace is sized for                  */
    int32_t dtype;           /* MI_DTYPE_*: operand type of the matrix-core GEMMs and attention  */
} mi_config;

/* Compute modes.  F32 is the parity mode (fp32 MFMA, exact fp32 arithmetic, <= 1e-4 of the CPU reference).
 * BF16 / F16 round the OPERANDS of every conv / linear GEMM and of the attention products to bf16 / fp16 and
 * feed them to v_mfma_f32_32x32x16_{bf16,f16}; accumulation, biases, GroupNorm / LayerNorm statistics, softmax,
 * GELU / GLU, the STFT / iSTFT and all activations in HBM stay float32 (BASELINE.json configs[2], configs[4]). */
#define MI_DTYPE_F32 0
#define MI_DTYPE_BF16 1
#define MI_DTYPE_F16 2

/* ---- model lifetime: replaces `states.load_model` + `model.to(device)`
 *      (demucs/states.py:50-80, demucs/apply.py:233) ------------------------------------- */
int mi_model_create(const mi_config *cfg, const mi_tensor_desc *weights, size_t n_weights, void **handle);
void mi_model_destroy(void *handle);

/* ---- `model(padded_mix)` under no_grad (demucs/apply.py:316-317 -> HTDemucs.forward,
 *      demucs/htdemucs.py:527-660).  mix_dev: (B, 2, segment_length); out_dev:
 *      (B, n_sources, 2, segment_length).  1 <= B <= max_batch. ---------------------------- */
int mi_model_forward(void *handle, const float *mix_dev, float *out_dev, int32_t B, void *stream);

/* ---- `HTDemucs.forward_core(mag, mix)` (demucs/htdemucs.py:662-759; the fork's ONNX "core", docs/onnx.md):
 *      the network without the STFT in front, the iSTFT behind and the branch sum.  mag_dev (B, 4, 2048, T) is the
 *      caller's `_magnitude(_spec(mix))` (the fork's ONNX / web tools compute it with their own STFT) and is what
 *      the frequency branch consumes; NULL = derive it from mix_dev (B, 2, segment_length) with the engine's STFT.
 *      spec_out_dev: (B, S, 4, 2048, T); time_out_dev: (B, S, 2, segment_length). --------------------------- */
int mi_model_forward_core(void *handle, const float *mix_dev, const float *mag_dev, float *spec_out_dev,
                          float *time_out_dev, int32_t B, void *stream);

/* ---- Hybrid Demucs v3 (`hdemucs_mmi`: demucs/hdemucs.py:338-794 with the constructor's defaults; BASELINE.json configs[4]).
 *      mi_config.segment_length is the LONGEST input the handle's workspace is sized for (remote/hdemucs_mmi.yaml: 44 s);
 *      every forward names its own length (the reference's HDemucs has no valid_length: demucs/apply.py:309-310 hands it
 *      each chunk as it is): mix_dev (B, 2, length) -> out_dev (B, n_sources, 2, length), 64 <= length <= segment_length.
 *      Tap names: "enc0".."enc5", "tenc0".."tenc4", "dec0+skip".."dec4+skip", "dec5", "tdec0+skip".."tdec3+skip", "tdec4"
 *      (decoder outputs are stored with the next layer's skip already added); frequency-branch taps carry a frame pitch of
 *      max(32, ceil(T / 4) * 4) floats per row. ------------------------------------------- */
int mi_hmodel_create(const mi_config *cfg, const mi_tensor_desc *weights, size_t n_weights, void **handle);
void mi_hmodel_destroy(void *handle);
int mi_hmodel_forward(void *handle, const float *mix_dev, float *out_dev, int32_t B, int32_t length, void *stream);
int mi_hmodel_tap(void *handle, const char *name, float *dst_dev, int32_t B, int64_t *numel_per_item, void *stream);
int64_t mi_hmodel_device_bytes(void *handle);
/* mi_hmodel_status: waits for `stream` and reports whether any forward of this handle so far lost its BLSTM recurrence
 *   (demucs/demucs.py:20-67 runs as one persistent kernel per sequence; a hidden-state wait that exceeds 0.3 s -- only when other
 *   persistent kernels keep part of its grid from becoming resident, e.g. two processes on one GPU -- abandons the sequence and sets a
 *   sticky word).  mi_hmodel_forward checks that word when it STARTS, so a time-out in the LAST forward of a job would go unnoticed:
 *   callers that hand results on (demucs_amd.apply.apply_model does) ask here once the work is enqueued.  MI_OK, or MI_EINVAL with the
 *   message of mi_last_error. */
int mi_hmodel_status(void *handle, void *stream);

/* Debug / parity aid: copy an internal activation left behind by the last mi_model_forward
 * (first B items) into dst_dev (may be NULL to query *numel_per_item only).  Names: "x0" (normalised CaC spectrogram),
 * "xt0", "enc0".."enc3", "tenc0".."tenc3" (encoder outputs = skip tensors; time-branch rows are stored with a
 * pitch rounded up to 4 samples), "tr_f", "tr_t"
 * (transformer outputs, channel-first), "yspec", "ytime" (decoder outputs before iSTFT). */
int mi_model_tap(void *handle, const char *name, float *dst_dev, int32_t B, int64_t *numel_per_item, void *stream);

/* Per-kernel-class device timing for the roofline report: between begin and end every launch of
 * the conv-GEMM and attention kernels made by mi_model_forward is bracketed by a HIP event pair
 * on the launch stream; end synchronises the stream and returns one row per kernel class with
 * the summed duration and the ALGORITHMIC flops / bytes of those launches. */
typedef struct mi_profile_row {
    char name[48];
    int64_t launches;
    double ms, flops, bytes;
} mi_profile_row;
/* mi_set_two_streams: the htdemucs engine runs the waveform branch of a forward on a side stream beside the spectral branch
 *   (enabled = 1, the default: results are bit-identical either way); 0 keeps every launch on the caller's stream, one kernel on
 *   the GPU at a time -- what a per-kernel timing wants.  Returns the previous setting.  Process-wide. */
int mi_set_two_streams(int32_t enabled);
/* mi_set_istft_fused: the inverse STFT (demucs/spec.py:30-47) runs its per-frame inverse transforms and the overlap-add in ONE
 *   kernel (hop blocks accumulated in registers; the windowed frames never go to memory).  0 selects the two separate kernels
 *   (frames to memory, then a gather): same summation order, bit-identical output -- kept for A/B runs and as the check of the
 *   fused kernel.  Process-wide; returns the previous setting.  Initial value: 1 unless MI_ISTFT_SPLIT is set. */
int mi_set_istft_fused(int32_t enabled);
/* mi_set_transpose_tiles: the kernels either side of the transforms (demucs/htdemucs.py:420-471, demucs/spec.py:11-47).  0, the
 *   default: the STFT walks a run of consecutive frames per workgroup (twiddles and window read once per run, the next frame's samples
 *   prefetched), and the layout changes between the frame-major scratch and the conv layout x[b][c][bin][frame] move 32 bins x all
 *   frames of a plane per workgroup (one contiguous run on the conv-layout side).  1 selects the kernels of rounds 1-3 for all three
 *   (one frame per workgroup, 32 x 32 tile transposes); as a bit mask 2 selects only the tile cac_transpose, 4 only the tile
 *   spec_transpose.  Same values, same arithmetic, bit-identical spectrograms and waveforms (the float64 normalisation sums are
 *   accumulated in another order) -- kept for A/B runs and as the check of the default kernels.  Process-wide; returns the previous
 *   setting.  Initial value: 0 unless MI_TRANSPOSE_TILES is set. */
int mi_set_transpose_tiles(int32_t enabled);
int mi_profile_begin(void *handle);
int mi_profile_end(void *handle, mi_profile_row *rows, int32_t max_rows, int32_t *n_rows, void *stream);

/* Device bytes held by the handle (weights + workspace). */
int64_t mi_model_device_bytes(void *handle);

/* ---- segment scheduler pieces (demucs/apply.py:257-301,108-124) --------------------------
 * Index arrays (`*_idx_dev`) are DEVICE arrays (tiny; e.g. torch int tensors) so that every call
 * stays asynchronous on `stream`.
 *
 * mi_segments_gather: TensorChunk.padded for a batch of segments.  For item i the segment is
 *   the `valid`-long window starting at track sample starts[i] (may be negative / run past the
 *   end: zero filled) of track_dev (channels, track_len); seg_dev is (B, channels, valid) and holds seg_capacity
 *   floats (checked: the call fails instead of writing past the buffer). */
int mi_segments_gather(const float *track_dev, int64_t track_len, int32_t channels, const int64_t *starts_idx_dev,
                       int32_t B, int32_t valid, float *seg_dev, int64_t seg_capacity, void *stream);

/* mi_ola_accumulate: `out[..., off:off+SL] += weight[:n] * chunk_out` for a batch
 *   (demucs/apply.py:295-296) with chunk_out = center_trim(model_out, n) (utils.py:38-54).
 *   model_out_dev: (B, rows, valid), rows = n_sources*channels; item i contributes samples
 *   [trim[i], trim[i]+lens[i]) of its rows, weighted by weight_dev[0:lens[i]], to
 *   acc_dev (rows, acc_len) at acc position offs[i].  [span_lo, span_hi) is the union of the
 *   items' acc ranges.  Items are applied in index order with separately rounded float32
 *   product and sum, i.e. exactly the reference's sequential loop.  model_out_dev holds out_capacity floats and
 *   weight_dev weight_len floats: the host-visible extents are checked, and because lens / trim live on the device
 *   the kernel itself never reads model_out_dev beyond `valid` per row nor weight_dev beyond weight_len. */
int mi_ola_accumulate(float *acc_dev, int64_t acc_len, int32_t rows, const float *model_out_dev, int32_t valid,
                      int64_t out_capacity, const int64_t *offs_idx_dev, const int32_t *lens_idx_dev,
                      const int32_t *trim_idx_dev, int32_t B, int64_t span_lo, int64_t span_hi, const float *weight_dev,
                      int32_t weight_len, void *stream);

/* mi_ola_finish: `out /= sum_weight` (demucs/apply.py:297-299); sum_weight is rebuilt from the
 *   (offs, lens) list of ALL segments of the track (sorted by offset, track coordinates) in the
 *   reference's float32 summation order.  In place on acc_dev (rows, acc_len); acc_off0 is the
 *   track position of acc sample 0; max_len = the segment length. */
int mi_ola_finish(float *acc_dev, int64_t acc_len, int32_t rows, int64_t acc_off0, const int64_t *offs_idx_dev,
                  const int32_t *lens_idx_dev, int32_t n_segments, int32_t max_len, const float *weight_dev, void *stream);

/* mi_resample_frac: `julius.resample_frac` as called by `demucs.audio.convert_audio` (demucs/audio.py:169-172), the step
 *   `Separator.separate_tensor` runs first when the input sample rate differs from the model's (demucs/api.py:265-266).
 *   old_sr / new_sr already divided by their gcd; table_dev (new_sr, 2*width + old_sr) is julius' windowed-sinc kernel bank
 *   (built by demucs_amd/audio.py); x_dev (rows, length) -> y_dev (rows, out_length), out_length <= new_sr*(length/old_sr + 1). */
int mi_resample_frac(const float *x_dev, int32_t rows, int64_t length, const float *table_dev, int32_t old_sr, int32_t new_sr,
                     int32_t width, float *y_dev, int64_t out_length, void *stream);

/* ---- track-level pre / post processing of `Separator` and the CLI, on the device -------------------------------------------
 * mi_mono_stats: `ref = wav.mean(0)`, then stats_dev[0] = ref.mean(), stats_dev[1] = ref.std() + 1e-8 (unbiased std;
 *   demucs/api.py:267-269), float32 results of float64 accumulation in a fixed order.  wav_dev (channels, length);
 *   scratch_dev: mi_mono_stats_scratch_bytes() bytes of device memory.  The scalars never visit the host.
 * mi_track_affine: in place on x_dev (numel floats): inverse = 0: `x -= mean; x /= std` (api.py:268-269), inverse = 1:
 *   `x *= std; x += mean` (api.py:285-288), each as two separately rounded float32 operations like the reference's. */
int32_t mi_mono_stats_scratch_bytes(void);
int mi_mono_stats(const float *wav_dev, int32_t channels, int64_t length, void *scratch_dev, float *stats_dev, void *stream);
int mi_track_affine(float *x_dev, int64_t numel, const float *stats_dev, int32_t inverse, void *stream);

/* mi_prevent_clip: `demucs.audio.prevent_clip(wav, mode)` (demucs/audio.py:218-234) on device stems: y = x / max(1.01 * max|x|, 1)
 *   ("rescale"; the peak is reduced on the device into peak_dev, 4 bytes), clamp(x, -0.99, 0.99) or tanh(x).
 * mi_two_stems: the `--two-stems` outputs of demucs/separate.py:189-218: minus = 0: y = 0 + (every stem but `selected`, in
 *   index order) -- the "no_STEM" file; minus = 1: y = origin - stems[selected] -- the "minus_STEM" file.  stems_dev is a
 *   HOST array of n_stems (<= 8) device pointers of numel floats each. */
#define MI_CLIP_RESCALE 1
#define MI_CLIP_CLAMP 2
#define MI_CLIP_TANH 3
int mi_prevent_clip(const float *x_dev, int64_t numel, int32_t mode, void *peak_dev, float *y_dev, void *stream);
int mi_two_stems(const float *const *stems_dev, int32_t n_stems, int32_t selected, const float *origin_dev, int32_t minus, int64_t numel,
                 float *y_dev, void *stream);

/* ---- kernel-level entry points (parity tests; same kernels the forward uses) --------------
 * mi_stft_cac: `_magnitude(_spec(mix))` (demucs/htdemucs.py:420-461, demucs/spec.py:11-27):
 *   mix_dev (B,2,L) -> cac_dev (B,4,2048,ceil(L/1024)), channel order [c0.re,c0.im,c1.re,c1.im]. */
int mi_stft_cac(const float *mix_dev, int32_t B, int32_t L, float *cac_dev, void *stream);
/* mi_istft_cac: `_ispec(_mask(z, x), length)` (demucs/htdemucs.py:442-471, demucs/spec.py:30-47):
 *   x_dev (B,S,4,2048,T) -> wav_dev (B,S,2,L) with T = ceil(L/1024). */
int mi_istft_cac(const float *x_dev, int32_t B, int32_t S, int32_t L, float *wav_dev, void *stream);

/* Generic convolution / linear layer evaluated by the implicit-GEMM MFMA kernel (stands in for
 *   F.conv1d / F.conv2d / F.conv_transpose / F.linear as invoked at demucs/hdemucs.py:110,116,
 *   136,153,287,294,313,326, demucs/demucs.py:138,140, demucs/htdemucs.py:589-599).
 *   See demucs_amd/csrc/gemm_conv.h (mi_conv_desc) for the field meanings. */
struct mi_conv_desc;
int mi_conv_forward(const struct mi_conv_desc *desc, void *stream);

/* Re-packs fp32 weights Wt[Kpad][Mpad] (the mi_conv_desc.wt layout) into the split-bf16 tile image
 *   [Mpad/tile_m][Kpad/16][3 terms][2 k-halves][tile_m][8] bf16 (6 * Kpad * Mpad bytes) that mi_conv_desc.wx takes:
 *   each weight w = hi + mid + lo exactly, so the 6-product bf16 MFMA loop reproduces the fp32 product sum
 *   (no reference counterpart: it is the load-time half of how F.conv / F.linear run on the matrix cores).
 *   tile_m must be the tile the layer will run with (64, 96 or 128). */
int mi_conv_pack_split(const float *wt_dev, int32_t Kpad, int32_t Mpad, int32_t tile_m, void *wx_dev, void *stream);

/* mi_conv_pack_tap: weights of a k x k stride-1 conv for the operand-image route of the half modes (gemm_tap.hip): from the
 *   packed float32 Wt[Kpad][Mpad] with k = ci * ntaps + tap to the 16-bit image Wtap[pairs][Mpad][8], pair =
 *   (ci / 8) * ntaps + tap, pairs rounded up to a multiple of 4: 16 * pairs * Mpad bytes.  Cin %% 8 == 0. */
int mi_conv_pack_tap(const float *wt_dev, int32_t Mpad, int32_t Cin, int32_t ntaps, int32_t dtype, void *wtap_dev, void *stream);
/* mi_f32_to_image: float32 x (B, C, P) (channel stride P) -> the 16-bit operand image [C / 8][B * P][8] a matrix product reads by
 *   LDS-DMA (half modes; used where the producer of x cannot write the image itself: the DConv branch's output in front of a
 *   decoder's transposed conv).  C %% 8 == 0. */
int mi_f32_to_image(const float *x_dev, int32_t B, int32_t C, int64_t P, int32_t dtype, void *img_dev, void *stream);

/* Converts fp32 weights Wt[Kpad][Mpad] (the mi_conv_desc.wt layout) into the bf16 / fp16 operand image
 *   Wh[ceil(Kpad/32)*4][Mpad][8] (2 * round_up(Kpad, 32) * Mpad bytes) that mi_conv_desc.wh takes when mi_conv_desc.half
 *   = MI_DTYPE_BF16 / MI_DTYPE_F16 (the load-time half of the reduced-precision compute modes; no reference counterpart). */
int mi_conv_pack_half(const float *wt_dev, int32_t Kpad, int32_t Mpad, int32_t dtype, void *wh_dev, void *stream);

/* Multi-head attention core softmax(QK^T/sqrt(64))V on channel-first tensors (stands in for the
 *   attention inside nn.MultiheadAttention, called at demucs/transformer.py:418-419,506):
 *   q_dev (B, heads*64, Tq), k_dev / v_dev (B, heads*64, Tk), tokens contiguous, with the given batch strides; o_dev like
 *   q_dev.  dtype = MI_DTYPE_F32 (fp32 MFMA, parity mode) or MI_DTYPE_BF16 / MI_DTYPE_F16 (operands of both products
 *   rounded to that type, float32 softmax and accumulation). */
int mi_attention(const float *q_dev, const float *k_dev, const float *v_dev, float *o_dev, int32_t B, int32_t heads,
                 int32_t Tq, int32_t Tk, int64_t q_batch_stride, int64_t kv_batch_stride, int64_t o_batch_stride,
                 int32_t dtype, void *stream);

/* mi_attention_heads: the attention core of the half modes on the operands the projections really write in those modes:
 *   q / k / v_dev are 16-bit (bf16 / fp16 by `dtype`) per-head token-major tensors [B][heads][T pitch][64] (what the
 *   MI_FLAG_HEADS epilogue of the in-projection produces; K / V tiles reach LDS by DMA); o_dev float32 (B, heads * 64, Tq)
 *   = softmax(q k^T / 8) v per head (demucs/transformer.py:339-377,466-512). */
int mi_attention_heads(const void *q_dev, const void *k_dev, const void *v_dev, float *o_dev, int32_t B, int32_t heads, int32_t Tq, int32_t Tk,
                       int32_t Tq_pitch, int32_t Tk_pitch, int32_t dtype, void *stream);

/* mi_attention in a half mode with the result written as the 16-bit operand image of the projection that consumes it
 *   (out_proj, demucs/transformer.py:418-419 inside nn.MultiheadAttention): img_dev[(heads * 64) / 8][n_img][8] bf16 / fp16,
 *   column b * Tq + query, n_img >= B * Tq; element values are the float32 results of mi_attention rounded to nearest even.
 *   Feeds mi_conv_desc.xh (demucs_amd/csrc/gemm_conv.h). */
int mi_attention_image(const float *q_dev, const float *k_dev, const float *v_dev, void *img_dev, int64_t n_img, int32_t B,
                       int32_t heads, int32_t Tq, int32_t Tk, int64_t q_batch_stride, int64_t kv_batch_stride, int32_t dtype,
                       void *stream);

/* In-place GroupNorm(1, C) + GELU of the first C channels of x (B, C_alloc, D1, D2) given per-row (mean, rstd) float2 statistics
 *   (row = b*D1 + d1 if row_mode else b): the norm/activation pair inside DConv (demucs/demucs.py:139). */
int mi_gn_gelu(float *x_dev, int32_t B, int32_t C, int32_t C_alloc, int32_t D1, int32_t D2, int32_t row_mode, const float *stats_dev,
               const float *w_dev, const float *b_dev, void *stream);

/* The same GroupNorm(1, h) + GELU in place on the valid columns of x (B, C_alloc, D1, pitch >= D2) AND, in the same pass, the
 *   Gram sums of the normalised hidden activations g per statistics row: gram_dev (rows x slots x HP x HP float64, zero on
 *   entry, HP = mi_gram_order(h)) receives G[i][j] += g_i g_j for i, j <= h in the upper 32 x 32 block triangle, with
 *   g_h = 1 (so G[i][h] = sum g_i).  They give the statistics of the GroupNorm(1, 2C) that follows DConv's 1x1 conv
 *   z = W g + b (demucs/demucs.py:141-142) without evaluating it: mi_gram_finalize computes per row
 *   sum z^2 = <wt, G> + cols * sum_bsq and sum z = ct . G[:, h] + cols * sum_b, writes (mean, rstd) float2 and re-zeroes
 *   gram_dev.  wt_dev: HP x HP float64 (W^T W on diagonal blocks, 2 W^T W above them, 2 W^T b in column h); ct_dev: HP
 *   float64 column sums of W. */
int32_t mi_gram_order(int32_t h);
int mi_gn_gelu_gram(float *x_dev, int32_t B, int32_t h, int32_t C_alloc, int32_t D1, int32_t D2, int32_t pitch, int32_t row_mode,
                    const float *stats_dev, const float *w_dev, const float *b_dev, double *gram_dev, int32_t slots, void *stream);
int mi_gram_finalize(double *gram_dev, int32_t rows, int32_t h, int32_t slots, const double *wt_dev, const double *ct_dev, double sum_b,
                     double sum_bsq, double cols, double count, float eps, float *stats_out_dev, void *stream);

/* The recurrence of one bidirectional nn.LSTM layer as the BLSTM of demucs/demucs.py:20-67 runs it (zero initial state, gate order
 *   i, f, g, o): gx_dev (N, 2 directions, 4H, W) = W_ih x_t + b_ih + b_hh for every step (a GEMM done before), whh_host = the HOST
 *   array (2, 4H, H) of weight_hh_l{k} / weight_hh_l{k}_reverse; out_dev (N, 2H, W): forward hidden states in channels [0, H), backward
 *   ones in [H, 2H).  H = 192 or 384 (hdemucs_mmi's layers 4 / 5).  mode 0: one launch per time step; mode 1: the persistent kernel
 *   the engine uses (hidden state exchanged between workgroups as tagged 8-byte granules, bounded waits).  Synchronous (test entry). */
int mi_lstm_seq(const float *gx_dev, const float *whh_host, int32_t N, int32_t H, int32_t W, float *out_dev, int32_t mode, void *stream);

/* LayerNorm over the channel axis of channel-first tokens x (B, C, T), optional additive table
 *   add_dev (C, T) (nn.LayerNorm at demucs/transformer.py:434-436,591-592 + positional
 *   embedding add :655-663). */
int mi_layernorm_cf(const float *x_dev, int32_t B, int32_t C, int32_t T, const float *w_dev, const float *b_dev,
                    const float *add_dev, float *y_dev, void *stream);

/* Debug aid: `hook(stream)` is called after every kernel launch the library makes (NULL switches it off).  Used by
 * tools/micro/poison_all.py to interleave a register / LDS poisoning kernel between the engine's kernels. */
void mi_debug_set_post_launch_hook(void (*hook)(void *stream));

/* Debug aid for the kernel tests: which main loop the LAST mi_conv_forward of this process took -- 0 table-driven gather /
 * register-staged loader (conv_gemm_kernel), 1 LDS-DMA plain linear tile (conv_gemm_dma_kernel), 2 LDS-DMA shifted-run taps
 * (conv_gemm_dmatap_kernel), 3 LDS-DMA row taps (conv_gemm_dmarow_kernel), 4 split-bf16 (gemm_x6.hip), 5 half-mode loops
 * (gemm_half.hip), 6 half-mode tap images (gemm_tap.hip); -1 before any call.  A route is chosen from the descriptor alone, so a
 * test that compares two routes bit for bit also has to see that they WERE two routes.  Process-wide, not thread-safe. */
int mi_debug_last_conv_route(void);

const char *mi_last_error(void);
/* "demucs_amd <version> gfx950" */
const char *mi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DEMUCS_AMD_H */
