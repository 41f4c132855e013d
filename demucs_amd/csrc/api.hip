// extern "C" surface of libdemucs_amd.so (declared in include/demucs_amd.h).
#include <new>

#include "common.h"
#include "gemm_conv.h"
#include "kernels.h"
#include "hmodel.h"
#include "model.h"

namespace mi {

post_launch_hook_t g_post_launch_hook = nullptr;
int g_last_conv_route = -1;

char *last_error_buf() {
    static thread_local char buf[1024] = "";
    return buf;
}

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 1024, fmt, ap);
    va_end(ap);
    return code;
}

// small cache of FFT tables for the handle-free kernel-level entry points
struct Tables {
    FftTables t{};
    bool ready = false;
};
static int get_tables(FftTables *out) {
    static Tables tb;
    if (!tb.ready) {
        std::vector<float> win(4096), env(1024);
        const std::vector<float2> tw = fft_twiddle_table();
        for (int i = 0; i < 4096; ++i) win[i] = 0.5f - 0.5f * cosf((float)i * (float)(2.0 * M_PI / 4096.0));
        for (int r = 0; r < 1024; ++r) { float e = 0.f; for (int j = 3; j >= 0; --j) e += win[r + 1024 * j] * win[r + 1024 * j]; env[r] = e; }
        float *dw, *de; float2 *dt;
        MI_HIP(hipMalloc((void **)&dw, 4096 * 4)); MI_HIP(hipMalloc((void **)&dt, tw.size() * 8)); MI_HIP(hipMalloc((void **)&de, 1024 * 4));
        MI_HIP(hipMemcpy(dw, win.data(), 4096 * 4, hipMemcpyHostToDevice));
        MI_HIP(hipMemcpy(dt, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
        MI_HIP(hipMemcpy(de, env.data(), 1024 * 4, hipMemcpyHostToDevice));
        tb.t = FftTables{dw, dt, de};
        tb.ready = true;
    }
    *out = tb.t;
    return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" {

const char *mi_last_error(void) { return last_error_buf(); }
void mi_debug_set_post_launch_hook(void (*hook)(void *stream)) { g_post_launch_hook = hook; }
int mi_debug_last_conv_route(void) { return g_last_conv_route; }
const char *mi_version(void) { return "demucs_amd 0.1 gfx950"; }

int mi_model_create(const mi_config *cfg, const mi_tensor_desc *weights, size_t n_weights, void **handle) {
    if (!cfg || !weights || !handle) return set_error(MI_EINVAL, "mi_model_create: null argument");
    *handle = nullptr;
    Model *m = new (std::nothrow) Model();
    if (!m) return set_error(MI_ENOMEM, "mi_model_create: host allocation failed");
    const int r = m->init(*cfg, weights, n_weights);
    if (r != MI_OK) { delete m; return r; }
    *handle = m;
    return MI_OK;
}

void mi_model_destroy(void *handle) {
    if (!handle) return;
    (void)hipDeviceSynchronize();
    delete (Model *)handle;
}

int mi_model_forward(void *handle, const float *mix_dev, float *out_dev, int32_t B, void *stream) {
    if (!handle) return set_error(MI_EINVAL, "mi_model_forward: null handle");
    return ((Model *)handle)->forward(mix_dev, out_dev, B, (hipStream_t)stream);
}

int mi_model_tap(void *handle, const char *name, float *dst_dev, int32_t B, int64_t *numel_per_item, void *stream) {
    if (!handle || !name || !numel_per_item) return set_error(MI_EINVAL, "mi_model_tap: null argument");
    const float *src = nullptr;
    const float **ptr_dev = &src;
    Model *m = (Model *)handle;
    const std::string n(name);
    const int64_t T = m->T, Tf = 8 * T, Tt = m->Lt[4];
    static const int ch[4] = {48, 96, 192, 384}, fr[4] = {512, 128, 32, 8};
    *ptr_dev = nullptr;
    if (n == "x0") { *ptr_dev = m->w_x0; *numel_per_item = 4 * 2048 * T; }
    else if (n == "xt0") { *ptr_dev = m->w_xt0; *numel_per_item = 2 * (int64_t)m->SL; }
    else if (n == "yspec") { *ptr_dev = m->w_yspec; *numel_per_item = 4 * (int64_t)m->S * 2048 * T; }
    else if (n == "ytime") { *ptr_dev = m->w_ytime; *numel_per_item = 2 * (int64_t)m->S * m->SL; }
    else if (n == "tr_f") { *ptr_dev = m->w_tr_x[0][1]; *numel_per_item = 512 * Tf; }   // 5 layers: ends in buffer 1
    else if (n == "tr_t") { *ptr_dev = m->w_tr_x[1][1]; *numel_per_item = 512 * Tt; }
    else if (n.size() == 4 && n.compare(0, 3, "enc") == 0 && n[3] >= '0' && n[3] <= '3') {
        const int i = n[3] - '0'; *ptr_dev = m->w_skip[i]; *numel_per_item = (int64_t)ch[i] * fr[i] * T;
    } else if (n.size() == 5 && n.compare(0, 4, "tenc") == 0 && n[4] >= '0' && n[4] <= '3') {
        const int i = n[4] - '0'; *ptr_dev = m->w_skip_t[i]; *numel_per_item = (int64_t)ch[i] * m->Lp[i + 1];   // rows padded to a multiple of 4
    }
    if (!*ptr_dev) return set_error(MI_EINVAL, "mi_model_tap: unknown tap '%s'", name);
    if (dst_dev) {
        MI_REQUIRE(B >= 1 && B <= m->cfg.max_batch, "mi_model_tap: batch %d out of range", B);
        MI_HIP(hipMemcpyAsync(dst_dev, src, sizeof(float) * (size_t)B * *numel_per_item, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return MI_OK;
}

int mi_model_forward_core(void *handle, const float *mix_dev, const float *mag_dev, float *spec_out_dev, float *time_out_dev,
                          int32_t B, void *stream) {
    if (!handle) return set_error(MI_EINVAL, "mi_model_forward_core: null handle");
    return ((Model *)handle)->forward_core(mix_dev, mag_dev, spec_out_dev, time_out_dev, B, (hipStream_t)stream);
}

// ---- Hybrid Demucs v3 (hdemucs_mmi) handles ------------------------------------------------------------------------
int mi_hmodel_create(const mi_config *cfg, const mi_tensor_desc *weights, size_t n_weights, void **handle) {
    if (!cfg || !weights || !handle) return set_error(MI_EINVAL, "mi_hmodel_create: null argument");
    *handle = nullptr;
    HModel *m = new (std::nothrow) HModel();
    if (!m) return set_error(MI_ENOMEM, "mi_hmodel_create: host allocation failed");
    const int r = m->hinit(*cfg, weights, n_weights);
    if (r != MI_OK) { delete m; return r; }
    *handle = m;
    return MI_OK;
}

void mi_hmodel_destroy(void *handle) {
    if (!handle) return;
    (void)hipDeviceSynchronize();
    delete (HModel *)handle;
}

int mi_hmodel_forward(void *handle, const float *mix_dev, float *out_dev, int32_t B, int32_t length, void *stream) {
    if (!handle) return set_error(MI_EINVAL, "mi_hmodel_forward: null handle");
    return ((HModel *)handle)->hforward(mix_dev, out_dev, B, length, (hipStream_t)stream);
}

int mi_hmodel_tap(void *handle, const char *name, float *dst_dev, int32_t B, int64_t *numel_per_item, void *stream) {
    if (!handle || !name || !numel_per_item) return set_error(MI_EINVAL, "mi_hmodel_tap: null argument");
    HModel *m = (HModel *)handle;
    auto it = m->taps.find(name);
    if (it == m->taps.end()) return set_error(MI_EINVAL, "mi_hmodel_tap: unknown tap '%s'", name);
    *numel_per_item = it->second.second;
    if (dst_dev) {
        MI_REQUIRE(B >= 1 && B <= m->cfg.max_batch, "mi_hmodel_tap: batch %d out of range", B);
        MI_HIP(hipMemcpyAsync(dst_dev, it->second.first, sizeof(float) * (size_t)B * it->second.second, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return MI_OK;
}

int mi_hmodel_status(void *handle, void *stream) {
    if (!handle) return set_error(MI_EINVAL, "mi_hmodel_status: null handle");
    MI_HIP(hipStreamSynchronize((hipStream_t)stream));
    HModel *m = (HModel *)handle;
    MI_REQUIRE(!m->lstm_timeout || *(volatile unsigned *)m->lstm_timeout == 0,
               "a forward's persistent LSTM kernel timed out waiting for its hidden-state exchange (GPU shared with another process's "
               "persistent kernels?): its output and every later one are invalid; MI_LSTM_STEPS=1 selects the one-launch-per-step recurrence");
    return MI_OK;
}

int64_t mi_hmodel_device_bytes(void *handle) { return handle ? ((HModel *)handle)->device_bytes + ((HModel *)handle)->hws_bytes : 0; }

int mi_set_two_streams(int32_t enabled) {
    const int old = g_two_streams;
    g_two_streams = enabled ? 1 : 0;
    return old;
}

int mi_set_istft_fused(int32_t enabled) {
    const int old = g_istft_fused;
    g_istft_fused = enabled ? 1 : 0;
    return old;
}

int mi_set_transpose_tiles(int32_t enabled) {
    const int old = g_transpose_tiles == 7 ? 1 : g_transpose_tiles;
    g_transpose_tiles = enabled == 1 ? 7 : (enabled & 7);      // 1 = all three round-3 kernels; 2 / 4 / 6: see kernels.h
    return old;
}

int mi_profile_begin(void *handle) {
    if (!handle) return set_error(MI_EINVAL, "mi_profile_begin: null handle");
    ((Model *)handle)->prof.begin();
    return MI_OK;
}

int mi_profile_end(void *handle, mi_profile_row *rows, int32_t max_rows, int32_t *n_rows, void *stream) {
    if (!handle || !rows || !n_rows) return set_error(MI_EINVAL, "mi_profile_end: null argument");
    Model *m = (Model *)handle;
    MI_TRY(m->prof.end((hipStream_t)stream));
    int n = 0;
    for (const ProfRow &r : m->prof.rows) {
        if (!r.launches || n >= max_rows) continue;
        mi_profile_row &o = rows[n++];
        memset(&o, 0, sizeof(o));
        strncpy(o.name, r.name, sizeof(o.name) - 1);
        o.launches = r.launches; o.ms = r.ms; o.flops = r.flops; o.bytes = r.bytes;
    }
    *n_rows = n;
    return MI_OK;
}

int64_t mi_model_device_bytes(void *handle) {
    if (!handle) return 0;
    Model *m = (Model *)handle;
    return m->device_bytes + (m->ws ? m->ws->bytes : 0);       // weights + the (possibly shared) workspace
}

int mi_segments_gather(const float *track_dev, int64_t track_len, int32_t channels, const int64_t *starts_dev, int32_t B,
                       int32_t valid, float *seg_dev, int64_t seg_capacity, void *stream) {
    MI_REQUIRE(track_dev && starts_dev && seg_dev && B > 0 && valid > 0 && channels > 0, "mi_segments_gather: bad argument");
    MI_REQUIRE((int64_t)B * channels * valid <= seg_capacity, "mi_segments_gather: %d x %d x %d floats do not fit seg_dev (%lld)", B,
               channels, valid, (long long)seg_capacity);
    return launch_segments_gather(track_dev, track_len, channels, starts_dev, B, valid, seg_dev, (hipStream_t)stream);
}

int mi_ola_accumulate(float *acc_dev, int64_t acc_len, int32_t rows, const float *model_out_dev, int32_t valid,
                      int64_t out_capacity, const int64_t *offs_dev, const int32_t *lens_dev, const int32_t *trim_dev, int32_t B,
                      int64_t span_lo, int64_t span_hi, const float *weight_dev, int32_t weight_len, void *stream) {
    MI_REQUIRE(acc_dev && model_out_dev && offs_dev && lens_dev && trim_dev && weight_dev && B > 0 && rows > 0 && valid > 0 &&
               weight_len > 0, "mi_ola_accumulate: bad argument");
    MI_REQUIRE((int64_t)B * rows * valid <= out_capacity, "mi_ola_accumulate: %d x %d x %d floats exceed model_out_dev (%lld)", B, rows,
               valid, (long long)out_capacity);
    MI_REQUIRE(span_hi <= acc_len, "mi_ola_accumulate: span end %lld past the accumulator (%lld)", (long long)span_hi, (long long)acc_len);
    return launch_ola_accumulate(acc_dev, acc_len, rows, model_out_dev, valid, offs_dev, lens_dev, trim_dev, B, span_lo, span_hi,
                                 weight_dev, weight_len, (hipStream_t)stream);
}

int mi_ola_finish(float *acc_dev, int64_t acc_len, int32_t rows, int64_t acc_off0, const int64_t *offs_dev,
                  const int32_t *lens_dev, int32_t n_segments, int32_t max_len, const float *weight_dev, void *stream) {
    MI_REQUIRE(acc_dev && offs_dev && lens_dev && weight_dev && n_segments > 0, "mi_ola_finish: bad argument");
    return launch_ola_finish(acc_dev, acc_len, rows, acc_off0, offs_dev, lens_dev, n_segments, max_len, weight_dev,
                             (hipStream_t)stream);
}

int32_t mi_mono_stats_scratch_bytes(void) { return post_stats_scratch_bytes(); }

int mi_mono_stats(const float *wav_dev, int32_t channels, int64_t length, void *scratch_dev, float *stats_dev, void *stream) {
    MI_REQUIRE(wav_dev && scratch_dev && stats_dev && channels >= 1 && length >= 1, "mi_mono_stats: bad argument");
    return launch_mono_stats(wav_dev, channels, length, (double *)scratch_dev, stats_dev, (hipStream_t)stream);
}

int mi_track_affine(float *x_dev, int64_t numel, const float *stats_dev, int32_t inverse, void *stream) {
    MI_REQUIRE(x_dev && stats_dev && numel >= 1 && (inverse == 0 || inverse == 1), "mi_track_affine: bad argument");
    return launch_track_affine(x_dev, numel, stats_dev, inverse, (hipStream_t)stream);
}

int mi_prevent_clip(const float *x_dev, int64_t numel, int32_t mode, void *peak_dev, float *y_dev, void *stream) {
    MI_REQUIRE(x_dev && y_dev && peak_dev && numel >= 1, "mi_prevent_clip: bad argument");
    MI_REQUIRE(mode >= MI_CLIP_RESCALE && mode <= MI_CLIP_TANH, "mi_prevent_clip: unknown mode %d", mode);
    return launch_prevent_clip(x_dev, numel, mode, (unsigned *)peak_dev, y_dev, (hipStream_t)stream);
}

int mi_two_stems(const float *const *stems_dev, int32_t n_stems, int32_t selected, const float *origin_dev, int32_t minus, int64_t numel,
                 float *y_dev, void *stream) {
    MI_REQUIRE(stems_dev && y_dev && n_stems >= 1 && n_stems <= 8 && selected >= 0 && selected < n_stems && numel >= 1,
               "mi_two_stems: bad argument");
    MI_REQUIRE(!minus || origin_dev, "mi_two_stems: the \"minus\" method needs the original mix");
    for (int k = 0; k < n_stems; ++k) MI_REQUIRE(stems_dev[k], "mi_two_stems: null stem %d", k);
    return launch_two_stems(stems_dev, n_stems, selected, origin_dev, minus ? 1 : 0, numel, y_dev, (hipStream_t)stream);
}

// Kernel-level entry points.  They allocate their scratch with hipMalloc and free it after a
// stream synchronise: convenient for parity tests, not meant for the hot loop.
int mi_stft_cac(const float *mix_dev, int32_t B, int32_t L, float *cac_dev, void *stream) {
    MI_REQUIRE(mix_dev && cac_dev && B > 0 && L > 4096, "mi_stft_cac: bad argument");
    FftTables tb;
    MI_TRY(get_tables(&tb));
    const int T = (L + 1023) / 1024;
    float *zt = nullptr; double *stats = nullptr;
    MI_HIP(hipMalloc((void **)&zt, (size_t)B * T * 4 * 2048 * 4));
    MI_HIP(hipMalloc((void **)&stats, sizeof(double) * 2 * kStatSlots * B));
    hipStream_t st = (hipStream_t)stream;
    int r = MI_OK;
    if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * kStatSlots * B, st) != hipSuccess) r = set_error(MI_EHIP, "memset failed");
    if (r == MI_OK) r = launch_stft_frames(mix_dev, B, L, tb, zt, stats, st);
    if (r == MI_OK) r = launch_cac_transpose(zt, B, T, nullptr, cac_dev, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(zt); (void)hipFree(stats);
    return r;
}

int mi_istft_cac(const float *x_dev, int32_t B, int32_t S, int32_t L, float *wav_dev, void *stream) {
    MI_REQUIRE(x_dev && wav_dev && B > 0 && S > 0 && L > 4096, "mi_istft_cac: bad argument");
    FftTables tb;
    MI_TRY(get_tables(&tb));
    const int T = (L + 1023) / 1024;
    float *yt = nullptr, *fr = nullptr;
    MI_HIP(hipMalloc((void **)&yt, (size_t)B * S * T * 4 * 2048 * 4));
    MI_HIP(hipMalloc((void **)&fr, (size_t)B * S * T * 2 * 4096 * 4));
    hipStream_t st = (hipStream_t)stream;
    const int r = launch_istft(x_dev, B, S, L, nullptr, nullptr, nullptr, tb, yt, fr, wav_dev, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(yt); (void)hipFree(fr);
    return r;
}

int mi_conv_forward(const struct mi_conv_desc *desc, void *stream) {
    MI_REQUIRE(desc, "mi_conv_forward: null descriptor");
    return launch_conv(*desc, (hipStream_t)stream);
}

int mi_resample_frac(const float *x_dev, int32_t rows, int64_t length, const float *table_dev, int32_t old_sr, int32_t new_sr,
                     int32_t width, float *y_dev, int64_t out_length, void *stream) {
    MI_REQUIRE(x_dev && table_dev && y_dev && old_sr > 0 && new_sr > 0 && width > 0, "mi_resample_frac: bad argument");
    return launch_resample_frac(x_dev, rows, length, table_dev, old_sr, new_sr, width, y_dev, out_length, (hipStream_t)stream);
}

int mi_conv_pack_split(const float *wt_dev, int32_t Kpad, int32_t Mpad, int32_t tile_m, void *wx_dev, void *stream) {
    MI_REQUIRE(wt_dev && wx_dev && Kpad > 0 && Mpad > 0 && conv_x6_supported(tile_m), "mi_conv_pack_split: bad argument");
    return launch_pack_split(wt_dev, Kpad, Mpad, tile_m, wx_dev, (hipStream_t)stream);
}

int mi_conv_pack_tap(const float *wt_dev, int32_t Mpad, int32_t Cin, int32_t ntaps, int32_t dtype, void *wtap_dev, void *stream) {
    MI_REQUIRE(wt_dev && wtap_dev && Mpad > 0 && Cin > 0 && ntaps > 0, "mi_conv_pack_tap: bad argument");
    return launch_pack_tap(wt_dev, Mpad, Cin, ntaps, dtype, wtap_dev, (hipStream_t)stream);
}

int mi_f32_to_image(const float *x_dev, int32_t B, int32_t C, int64_t P, int32_t dtype, void *img_dev, void *stream) {
    MI_REQUIRE(x_dev && img_dev && B > 0 && C > 0 && P > 0, "mi_f32_to_image: bad argument");
    return launch_f32_to_image(x_dev, B, C, P, dtype, img_dev, (hipStream_t)stream);
}

int mi_conv_pack_half(const float *wt_dev, int32_t Kpad, int32_t Mpad, int32_t dtype, void *wh_dev, void *stream) {
    MI_REQUIRE(wt_dev && wh_dev && Kpad > 0 && Mpad > 0, "mi_conv_pack_half: bad argument");
    return launch_pack_half(wt_dev, Kpad, Mpad, dtype, wh_dev, (hipStream_t)stream);
}

int mi_attention(const float *q_dev, const float *k_dev, const float *v_dev, float *o_dev, int32_t B, int32_t heads, int32_t Tq,
                 int32_t Tk, int64_t q_batch_stride, int64_t kv_batch_stride, int64_t o_batch_stride, int32_t dtype, void *stream) {
    MI_REQUIRE(q_dev && k_dev && v_dev && o_dev && B > 0 && heads > 0 && Tq > 0 && Tk > 0, "mi_attention: bad argument");
    return launch_attention(q_dev, k_dev, v_dev, o_dev, B, heads, Tq, Tk, q_batch_stride, kv_batch_stride, o_batch_stride, dtype,
                            (hipStream_t)stream);
}

int mi_attention_heads(const void *q_dev, const void *k_dev, const void *v_dev, float *o_dev, int32_t B, int32_t heads, int32_t Tq, int32_t Tk,
                       int32_t Tq_pitch, int32_t Tk_pitch, int32_t dtype, void *stream) {
    MI_REQUIRE(q_dev && k_dev && v_dev && o_dev && B > 0 && heads > 0 && Tq > 0 && Tk > 0, "mi_attention_heads: bad argument");
    const void *zero = conv_zero_page();
    MI_REQUIRE(zero, "mi_attention_heads: could not allocate the zero page");
    return launch_attention_heads(q_dev, k_dev, v_dev, zero, B, heads, Tq, Tk, Tq_pitch, Tk_pitch, dtype, nullptr, 0, o_dev,
                                  (int64_t)heads * 64 * Tq, (hipStream_t)stream);
}

int mi_attention_image(const float *q_dev, const float *k_dev, const float *v_dev, void *img_dev, int64_t n_img, int32_t B,
                       int32_t heads, int32_t Tq, int32_t Tk, int64_t q_batch_stride, int64_t kv_batch_stride, int32_t dtype,
                       void *stream) {
    MI_REQUIRE(q_dev && k_dev && v_dev && img_dev && B > 0 && heads > 0 && Tq > 0 && Tk > 0, "mi_attention_image: bad argument");
    MI_REQUIRE(dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16, "mi_attention_image: dtype %d is not a half mode", dtype);
    return launch_attention(q_dev, k_dev, v_dev, nullptr, B, heads, Tq, Tk, q_batch_stride, kv_batch_stride, 0, dtype,
                            (hipStream_t)stream, img_dev, n_img);
}

int mi_gn_gelu(float *x_dev, int32_t B, int32_t C, int32_t C_alloc, int32_t D1, int32_t D2, int32_t row_mode, const float *stats_dev,
               const float *w_dev, const float *b_dev, void *stream) {
    MI_REQUIRE(x_dev && stats_dev && w_dev && b_dev && B > 0 && C > 0 && D1 > 0 && D2 > 0, "mi_gn_gelu: bad argument");
    MI_REQUIRE(C_alloc >= C, "mi_gn_gelu: C_alloc < C");
    return launch_gn_gelu(x_dev, B, C, C_alloc, D1, D2, row_mode, (const float2 *)stats_dev, w_dev, b_dev, (hipStream_t)stream);
}

int32_t mi_gram_order(int32_t h) { return gram_hp(h); }

int mi_lstm_seq(const float *gx_dev, const float *whh_host, int32_t N, int32_t H, int32_t W, float *out_dev, int32_t mode, void *stream) {
    MI_REQUIRE(gx_dev && whh_host && out_dev && N > 0 && W > 0 && (H == 192 || H == 384), "mi_lstm_seq: bad argument (H must be 192 or 384)");
    MI_REQUIRE(mode == 0 || mode == 1, "mi_lstm_seq: mode 0 (one launch per step) or 1 (persistent kernel)");
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> packed((size_t)2 * 4 * H * H);
    pack_lstm_whh(whh_host, H, packed.data());
    float *wd = nullptr, *state = nullptr;
    void *scratch = nullptr;
    unsigned *flag = nullptr;
    int r = MI_OK;
    auto fail = [&](hipError_t e, const char *what) { if (e != hipSuccess && r == MI_OK) r = set_error(MI_EHIP, "mi_lstm_seq: %s: %s", what, hipGetErrorString(e)); };
    fail(hipMalloc((void **)&wd, packed.size() * sizeof(float)), "hipMalloc");
    fail(hipMalloc((void **)&state, (size_t)6 * N * H * sizeof(float)), "hipMalloc");
    fail(hipMalloc(&scratch, lstm_persist_scratch_bytes()), "hipMalloc");
    fail(hipHostMalloc((void **)&flag, 64, hipHostMallocMapped), "hipHostMalloc");
    if (r == MI_OK) {
        *flag = 0;
        fail(hipMemcpyAsync(wd, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, st), "hipMemcpyAsync");
        if (r == MI_OK) r = mode ? launch_lstm_persist(gx_dev, wd, N, H, W, out_dev, scratch, flag, st) : launch_lstm_seq(gx_dev, wd, N, H, W, out_dev, state, st);
        fail(hipStreamSynchronize(st), "hipStreamSynchronize");
        if (r == MI_OK && *(volatile unsigned *)flag) r = set_error(MI_EHIP, "mi_lstm_seq: the persistent kernel timed out waiting for its hidden-state exchange");
        if (r == MI_OK && mode && getenv("MI_LSTM_DEBUG")) {
                        unsigned dbg[16] = {};
            (void)hipMemcpy(dbg, (char *)scratch + lstm_persist_ctl_offset(), sizeof(dbg), hipMemcpyDeviceToHost);
            fprintf(stderr, "[lstm] H %d N %d W %d: block 0 / wave 1 spent %.1f us gathering h in %u poll passes (%.2f us, %.2f passes per step); "
                    "%s stores\n", H, N, W, dbg[2] * 0.01, dbg[3], dbg[2] * 0.01 / std::max(1, W - 1), (double)dbg[3] / std::max(1, W - 1),
                    dbg[4] ? "same-XCD plain" : "write-through");
            fprintf(stderr, "[lstm]   shader cycles per step: gather %.0f, issue + products %.0f, LDS write + barrier %.0f, gate stage %.0f; whole step %.0f\n",
                    (double)dbg[8] / W, (double)dbg[9] / W, (double)dbg[10] / W, (double)dbg[11] / W, (double)dbg[12] / W);
        }
    }
    if (wd) (void)hipFree(wd);
    if (state) (void)hipFree(state);
    if (scratch) (void)hipFree(scratch);
    if (flag) (void)hipHostFree(flag);
    return r;
}

int mi_gn_gelu_gram(float *x_dev, int32_t B, int32_t h, int32_t C_alloc, int32_t D1, int32_t D2, int32_t pitch, int32_t row_mode,
                    const float *stats_dev, const float *w_dev, const float *b_dev, double *gram_dev, int32_t slots, void *stream) {
    MI_REQUIRE(x_dev && stats_dev && w_dev && b_dev && gram_dev && B > 0 && h > 0 && D1 > 0 && D2 > 0 && slots > 0, "mi_gn_gelu_gram: bad argument");
    MI_REQUIRE(C_alloc >= h && pitch >= D2, "mi_gn_gelu_gram: C_alloc < h or pitch < D2");
    return launch_gn_gelu_gram(x_dev, B, h, C_alloc, D1, D2, pitch, row_mode, (const float2 *)stats_dev, w_dev, b_dev, gram_dev, slots,
                               (hipStream_t)stream);
}

int mi_gram_finalize(double *gram_dev, int32_t rows, int32_t h, int32_t slots, const double *wt_dev, const double *ct_dev, double sum_b,
                     double sum_bsq, double cols, double count, float eps, float *stats_out_dev, void *stream) {
    MI_REQUIRE(gram_dev && wt_dev && ct_dev && stats_out_dev && rows > 0 && h > 0 && slots > 0 && count > 0.0, "mi_gram_finalize: bad argument");
    return launch_gram_finalize(gram_dev, rows, h, slots, wt_dev, ct_dev, sum_b, sum_bsq, cols, count, eps, (float2 *)stats_out_dev,
                                (hipStream_t)stream);
}

int mi_layernorm_cf(const float *x_dev, int32_t B, int32_t C, int32_t T, const float *w_dev, const float *b_dev,
                    const float *add_dev, float *y_dev, void *stream) {
    MI_REQUIRE(x_dev && w_dev && b_dev && y_dev, "mi_layernorm_cf: null argument");
    return launch_layernorm_cf(x_dev, B, C, T, w_dev, b_dev, add_dev, y_dev, nullptr, (hipStream_t)stream);
}

}  // extern "C"
