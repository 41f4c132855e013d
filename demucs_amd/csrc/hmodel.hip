// Hybrid Demucs v3 forward on gfx950 (`hdemucs_mmi` architecture; reference: demucs/hdemucs.py:689-794 in eval mode with
// the default hyper-parameters of demucs/hdemucs.py:366-410: depth 6, channels 48, cac, hybrid, dconv_mode 1 / comp 4,
// BLSTM + LocalState and GroupNorm(4) from layer 4).  ANY input length >= kMinLength: the reference's HDemucs has no
// `valid_length`, so `apply_model` hands it every chunk at its own length (demucs/apply.py:309-310).
//
// Layouts as in model.hip: channel-first, position axis contiguous; frequency branch x[b][C][Fr][T] (row pitch T), time
// branch xt[b][C][Lp] (pitch rounded up to 4).  Layers 0-3 run through the same implicit-GEMM kernels as htdemucs (DConv
// hidden width C/4 here); layers 4-5 (Fr = 1) add: GroupNorm(4) passes (row statistics + fused apply), the BLSTM
// (chunk unfold -> input-gate GEMMs -> one persistent workgroup per sequence and direction -> re-stitch), LocalState
// attention, and transposed convolutions whose GroupNorm statistics cover the UN-cropped output (hdemucs.py:325-333).
#include <algorithm>
#include <cmath>

#include "hmodel.h"
#include "kernels.h"

namespace mi {

constexpr int kMinLength = 1;          // the reference forwards chunks down to ONE sample (pad1d's zero-then-reflect rule); frequency-
                                       // branch rows carry a pitch >= 32 frames (the row-statistics epilogues reduce 32 columns at a time)
constexpr size_t kMaxGeos = 24;        // cached input lengths per handle (least recently used one evicted)
static const int hFr[5] = {2048, 512, 128, 32, 8};

int HModel::halloc(void **p, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return set_error(MI_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    hws.push_back(*p);
    hws_bytes += (int64_t)bytes;
    return MI_OK;
}
HModel::~HModel() {
    for (void *p : hws) (void)hipFree(p);
    if (lstm_timeout) (void)hipHostFree(lstm_timeout);
}

// MI_LSTM_STEPS=1: the round-3 recurrence (one launch per time step, hkernels.hip) instead of the persistent kernel of lstm.hip
static bool lstm_step_chain() {
    static const bool on = getenv("MI_LSTM_STEPS") != nullptr && atoi(getenv("MI_LSTM_STEPS")) != 0;
    return on;
}

int HModel::load_norm(const WeightTable &wt, const std::string &name, int C, float **w, float **b) {
    const float *pw, *pb;
    MI_TRY(wt.get(name + ".weight", C, &pw)); MI_TRY(wt.get(name + ".bias", C, &pb));
    MI_TRY(pack_vec(pw, C, C, false, w));
    return pack_vec(pb, C, C, false, b);
}

int HModel::load_deep(const WeightTable &wt, const std::string &prefix, int C, HDeepLayerW *l, int d) {
    const int H = C / 4;
    const std::string p = prefix + ".dconv.layers." + std::to_string(d);
    const float *w, *b;
    MI_TRY(wt.get(p + ".0.weight", (int64_t)H * C * 3, &w)); MI_TRY(wt.get(p + ".0.bias", H, &b));
    MI_TRY(pack_conv(w, b, H, 3 * C, false, &l->conv3, 3));         // half modes: + the tap-ordered image
    MI_TRY(load_norm(wt, p + ".1", H, &l->g1w, &l->g1b));
    for (int layer = 0; layer < 2; ++layer) {          // nn.LSTM(bidirectional, 2 layers): gate rows i, f, g, o
        const int Kin = layer ? 2 * H : H;
        std::vector<float> wih((size_t)8 * H * Kin), bih(8 * H), whh((size_t)2 * H * 4 * H);
        for (int dir = 0; dir < 2; ++dir) {
            const std::string sfx = "_l" + std::to_string(layer) + (dir ? "_reverse" : "");
            const float *a, *hh, *b1, *b2;
            MI_TRY(wt.get(p + ".3.lstm.weight_ih" + sfx, (int64_t)4 * H * Kin, &a));
            MI_TRY(wt.get(p + ".3.lstm.weight_hh" + sfx, (int64_t)4 * H * H, &hh));
            MI_TRY(wt.get(p + ".3.lstm.bias_ih" + sfx, 4 * H, &b1));
            MI_TRY(wt.get(p + ".3.lstm.bias_hh" + sfx, 4 * H, &b2));
            memcpy(&wih[(size_t)dir * 4 * H * Kin], a, sizeof(float) * 4 * H * Kin);
            for (int r = 0; r < 4 * H; ++r) {
                bih[dir * 4 * H + r] = b1[r] + b2[r];
                for (int k = 0; k < H; ++k) whh[((size_t)dir * 4 * H + r) * H + k] = hh[(size_t)r * H + k];
            }
        }
        MI_TRY(pack_conv(wih.data(), bih.data(), 8 * H, Kin, false, &l->ih[layer]));
        if (H == 192 || H == 384) {          // the matrix-pipe kernels' operand order
            std::vector<float> packed(whh.size());
            pack_lstm_whh(whh.data(), H, packed.data());
            MI_TRY(upload(packed, &l->whhT[layer]));
        } else MI_TRY(upload(whh, &l->whhT[layer]));     // small widths (demucs_unittest): natural order for lstm_small_kernel
    }
    MI_TRY(wt.get(p + ".3.linear.weight", (int64_t)H * 2 * H, &w)); MI_TRY(wt.get(p + ".3.linear.bias", H, &b));
    MI_TRY(pack_conv(w, b, H, 2 * H, false, &l->lin));
    {   // LocalState: one projection GEMM with rows [query; key; content; query_decay]
        const char *names[4] = {"query", "key", "content", "query_decay"};
        const int rows[4] = {H, H, H, 16};
        const int M = 3 * H + 16;
        std::vector<float> wq((size_t)M * H), bq(M);
        int r0 = 0;
        for (int q = 0; q < 4; ++q) {
            MI_TRY(wt.get(p + ".4." + names[q] + ".weight", (int64_t)rows[q] * H, &w));
            MI_TRY(wt.get(p + ".4." + names[q] + ".bias", rows[q], &b));
            memcpy(&wq[(size_t)r0 * H], w, sizeof(float) * rows[q] * H);
            memcpy(&bq[r0], b, sizeof(float) * rows[q]);
            r0 += rows[q];
        }
        MI_TRY(pack_conv(wq.data(), bq.data(), M, H, false, &l->qkc));
        MI_TRY(wt.get(p + ".4.proj.weight", (int64_t)H * H, &w)); MI_TRY(wt.get(p + ".4.proj.bias", H, &b));
        MI_TRY(pack_conv(w, b, H, H, false, &l->proj));
    }
    MI_TRY(wt.get(p + ".5.weight", (int64_t)2 * C * H, &w)); MI_TRY(wt.get(p + ".5.bias", 2 * C, &b));
    MI_TRY(pack_conv(w, b, 2 * C, H, false, &l->conv1));
    MI_TRY(load_norm(wt, p + ".6", 2 * C, &l->g2w, &l->g2b));
    MI_TRY(wt.get(p + ".8.scale", C, &w));
    return pack_vec(w, C, C, false, &l->ls);
}

int HModel::hinit(const mi_config &c, const mi_tensor_desc *weights, size_t n) {
    cfg = c;
    MI_REQUIRE(c.n_sources >= 1 && c.n_sources <= 8, "n_sources %d unsupported", c.n_sources);
    MI_REQUIRE(c.max_batch >= 1 && c.max_batch <= 16, "max_batch %d out of range [1, 16]", c.max_batch);
    MI_REQUIRE(c.segment_length >= kMinLength, "max length %d below the minimum of %d samples", c.segment_length, kMinLength);
    MI_REQUIRE(c.dtype == MI_DTYPE_F32 || c.dtype == MI_DTYPE_BF16 || c.dtype == MI_DTYPE_F16, "unknown compute dtype %d", c.dtype);
    S = c.n_sources; Lmax = c.segment_length; Tmax = (Lmax + 1023) / 1024;
    WeightTable wt;
    for (size_t i = 0; i < n; ++i) wt.t[weights[i].name] = {weights[i].data, weights[i].numel};
    {   // layer widths: `channels` of the reference constructor (hdemucs.py:366) = rows of encoder.0's bias; 48 for hdemucs_mmi, 4 for
        // the reference's own `demucs_unittest` model (pretrained.py:27-29); growth 2 per layer
        auto it = wt.t.find("encoder.0.conv.bias");
        MI_REQUIRE(it != wt.t.end(), "missing tensor 'encoder.0.conv.bias'");
        const int64_t c0 = it->second.second;
        MI_REQUIRE(c0 == 48 || c0 == 4, "HDemucs channels = %lld: the engine runs 48 (hdemucs_mmi) and 4 (demucs_unittest); other widths have "
                   "neither the matrix-pipe nor the generic recurrence / attention kernels", (long long)c0);
        for (int i = 0; i < 7; ++i) hCh[i] = (int)c0 << i;
    }
    const int C0 = hCh[0], C3 = hCh[3], C4 = hCh[4], C5 = hCh[5], C6 = hCh[6];
    {   // FFT tables (as in Model::init)
        std::vector<float> win(4096), env(1024);
        const std::vector<float2> tw = fft_twiddle_table();
        for (int i = 0; i < 4096; ++i) win[i] = 0.5f - 0.5f * cosf((float)i * (float)(2.0 * M_PI / 4096.0));
        for (int r = 0; r < 1024; ++r) { float e = 0.f; for (int j = 3; j >= 0; --j) e += win[r + 1024 * j] * win[r + 1024 * j]; env[r] = e; }
        float *dw, *de; float2 *dt;
        MI_TRY(upload(win, &dw)); MI_TRY(upload(tw, &dt)); MI_TRY(upload(env, &de));
        fft = FftTables{dw, dt, de};
    }
    const float *w, *b, *rw, *rb;
    for (int i = 0; i < 4; ++i) {                         // layers 0-3: both branches, htdemucs-shaped (DConv hidden C/4)
        const int Cin = i ? hCh[i - 1] : 4, Cint = i ? hCh[i - 1] : 2, C = hCh[i];
        const std::string p = "encoder." + std::to_string(i), pt = "tencoder." + std::to_string(i);
        MI_TRY(wt.get(p + ".conv.weight", (int64_t)C * Cin * 8, &w)); MI_TRY(wt.get(p + ".conv.bias", C, &b));
        MI_TRY(wt.get(p + ".rewrite.weight", (int64_t)2 * C * C, &rw)); MI_TRY(wt.get(p + ".rewrite.bias", 2 * C, &rb));
        MI_TRY(pack_conv(w, b, C, Cin * 8, false, &henc[i].conv));
        if (i) MI_TRY(pack_enc_tap(w, Cin, &henc[i].conv));        // half modes: stride-1 two-tap form over a phase-split image
        MI_TRY(pack_conv(rw, rb, 2 * C, C, true, &henc[i].rewrite));
        MI_TRY(load_dconv(wt, p, C, 1, 1, true, &henc[i].dconv, 4));
        MI_TRY(wt.get(pt + ".conv.weight", (int64_t)C * Cint * 8, &w)); MI_TRY(wt.get(pt + ".conv.bias", C, &b));
        MI_TRY(wt.get(pt + ".rewrite.weight", (int64_t)2 * C * C, &rw)); MI_TRY(wt.get(pt + ".rewrite.bias", 2 * C, &rb));
        MI_TRY(pack_conv(w, b, C, Cint * 8, false, &htenc[i].conv));
        if (i) MI_TRY(pack_enc_tap(w, Cint, &htenc[i].conv));
        MI_TRY(pack_conv(rw, rb, 2 * C, C, true, &htenc[i].rewrite));
        MI_TRY(load_dconv(wt, pt, C, 1, 1, false, &htenc[i].dconv, 4));
    }
    {   // frequency embedding 0.2 * (10 * weight).t() -> [48][512]
        const float *ew;
        MI_TRY(wt.get("freq_emb.embedding.weight", (int64_t)512 * C0, &ew));
        std::vector<float> emb((size_t)C0 * 512);
        for (int f = 0; f < 512; ++f)
            for (int c = 0; c < C0; ++c) emb[(size_t)c * 512 + f] = 0.2f * (ew[(size_t)f * C0 + c] * 10.0f);
        MI_TRY(upload(emb, &freq_emb));
    }
    // tencoder.4: the "empty" layer (conv only) whose output is injected into encoder.4
    MI_TRY(wt.get("tencoder.4.conv.weight", (int64_t)C4 * C3 * 8, &w)); MI_TRY(wt.get("tencoder.4.conv.bias", C4, &b));
    MI_TRY(pack_conv(w, b, C4, C3 * 8, false, &htenc[4].conv));
    // encoder.4 (last frequency layer, Fr 8 -> 1) and encoder.5 (time layer k = 4, s = 2): GroupNorm(4), deep DConv
    for (int i = 4; i < 6; ++i) {
        const int Cin = i == 4 ? C3 : C4, C = 2 * Cin, ker = i == 4 ? 8 : 4;
        const std::string p = "encoder." + std::to_string(i);
        MI_TRY(wt.get(p + ".conv.weight", (int64_t)C * Cin * ker, &w)); MI_TRY(wt.get(p + ".conv.bias", C, &b));
        MI_TRY(pack_conv(w, b, C, Cin * ker, false, &henc[i].conv));
        MI_TRY(load_norm(wt, p + ".norm1", C, &henc[i].n1w, &henc[i].n1b));
        MI_TRY(wt.get(p + ".rewrite.weight", (int64_t)2 * C * C, &rw)); MI_TRY(wt.get(p + ".rewrite.bias", 2 * C, &rb));
        MI_TRY(pack_conv(rw, rb, 2 * C, C, false, &henc[i].rewrite));            // natural row order: GroupNorm(4) before the GLU
        MI_TRY(load_norm(wt, p + ".norm2", 2 * C, &henc[i].n2w, &henc[i].n2b));
        for (int d = 0; d < 2; ++d) MI_TRY(load_deep(wt, p, C, &henc[i].deep[d], d));
    }
    // decoder.0 (mirror of encoder.5): Conv1d k3 -> GN -> GLU -> ConvTranspose1d(k4, s2) -> GN -> crop -> GELU
    MI_TRY(wt.get("decoder.0.rewrite.weight", (int64_t)C6 * C5 * 3, &rw)); MI_TRY(wt.get("decoder.0.rewrite.bias", C6, &rb));
    MI_TRY(pack_conv(rw, rb, C6, C5 * 3, false, &hdec[0].rewrite, 3));
    MI_TRY(load_norm(wt, "decoder.0.norm1", C6, &hdec[0].n1w, &hdec[0].n1b));
    MI_TRY(wt.get("decoder.0.conv_tr.weight", (int64_t)C5 * C4 * 4, &w)); MI_TRY(wt.get("decoder.0.conv_tr.bias", C4, &b));
    MI_TRY(pack_convtr(w, b, C5, C4, &hdec[0].convtr, 2));
    MI_TRY(load_norm(wt, "decoder.0.norm2", C4, &hdec[0].n2w, &hdec[0].n2b));
    {   // decoder.1 (mirror of encoder.4, Fr = 1): of the 3x3 rewrite only the middle frequency row ever meets data
        MI_TRY(wt.get("decoder.1.rewrite.weight", (int64_t)C5 * C4 * 9, &rw)); MI_TRY(wt.get("decoder.1.rewrite.bias", C5, &rb));
        std::vector<float> mid((size_t)C5 * C4 * 3);
        for (size_t mc = 0; mc < (size_t)C5 * C4; ++mc)
            for (int k2 = 0; k2 < 3; ++k2) mid[mc * 3 + k2] = rw[mc * 9 + 3 + k2];
        MI_TRY(pack_conv(mid.data(), rb, C5, C4 * 3, false, &hdec[1].rewrite, 3));
        MI_TRY(load_norm(wt, "decoder.1.norm1", C5, &hdec[1].n1w, &hdec[1].n1b));
        MI_TRY(wt.get("decoder.1.conv_tr.weight", (int64_t)C4 * C3 * 8, &w)); MI_TRY(wt.get("decoder.1.conv_tr.bias", C3, &b));
        MI_TRY(pack_convtr(w, b, C4, C3, &hdec[1].convtr, 4));
        MI_TRY(load_norm(wt, "decoder.1.norm2", C3, &hdec[1].n2w, &hdec[1].n2b));
        MI_TRY(wt.get("tdecoder.0.conv_tr.weight", (int64_t)C4 * C3 * 8, &w)); MI_TRY(wt.get("tdecoder.0.conv_tr.bias", C3, &b));
        MI_TRY(pack_convtr(w, b, C4, C3, &htdec[0].convtr, 4));
        MI_TRY(load_norm(wt, "tdecoder.0.norm2", C3, &htdec[0].n2w, &htdec[0].n2b));
    }
    for (int j = 2; j < 6; ++j) {                         // decoder.2-5 / tdecoder.1-4: htdemucs-shaped, no DConv (dconv_mode = 1)
        const int i = 5 - j, C = hCh[i];
        const int Cout = i ? hCh[i - 1] : 4 * S, Coutt = i ? hCh[i - 1] : 2 * S;
        const std::string p = "decoder." + std::to_string(j), pt = "tdecoder." + std::to_string(j - 1);
        MI_TRY(wt.get(p + ".rewrite.weight", (int64_t)2 * C * C * 9, &rw)); MI_TRY(wt.get(p + ".rewrite.bias", 2 * C, &rb));
        MI_TRY(pack_conv(rw, rb, 2 * C, C * 9, true, &hdec[j].rewrite, 9));     // half modes: + the tap-ordered image (gemm_tap.hip)
        MI_TRY(wt.get(p + ".conv_tr.weight", (int64_t)C * Cout * 8, &w)); MI_TRY(wt.get(p + ".conv_tr.bias", Cout, &b));
        MI_TRY(pack_convtr(w, b, C, Cout, &hdec[j].convtr, 4));
        MI_TRY(wt.get(pt + ".rewrite.weight", (int64_t)2 * C * C * 3, &rw)); MI_TRY(wt.get(pt + ".rewrite.bias", 2 * C, &rb));
        MI_TRY(pack_conv(rw, rb, 2 * C, C * 3, true, &htdec[j - 1].rewrite, 3));
        MI_TRY(wt.get(pt + ".conv_tr.weight", (int64_t)C * Coutt * 8, &w)); MI_TRY(wt.get(pt + ".conv_tr.bias", Coutt, &b));
        MI_TRY(pack_convtr(w, b, C, Coutt, &htdec[j - 1].convtr, 4));
    }

    // ---- workspace for max_batch items of Lmax samples -------------------------------------------------------------
    const size_t B = c.max_batch, Tx = Tmax, T = std::max<size_t>(32, (Tmax + 3) / 4 * 4), T5 = (Tx + 1) / 2;     // T: padded frame pitch
    int lt[6], lp[6];
    lt[0] = Lmax;
    for (int i = 0; i < 5; ++i) lt[i + 1] = (lt[i] + 3) / 4;
    for (int i = 0; i < 6; ++i) lp[i] = round_up(lt[i], 4);
    auto A = [&](float **p, size_t per_item) { return halloc((void **)p, (per_item + 64) * B * sizeof(float)); };
    MI_TRY(A(&x_t0, (size_t)2 * lp[0])); MI_TRY(A(&x_zt, 4 * 2048 * T)); MI_TRY(A(&x_0, 4 * 2048 * T));
    size_t big = 0, hid = 0;          // largest layer tensor; largest DConv hidden tensor (round_up(C / 4, 16) channels: more than C when C < 16)
    for (int i = 0; i < 4; ++i) {
        const size_t nf = (size_t)hCh[i] * hFr[i + 1] * T, nt = (size_t)hCh[i] * lp[i + 1];
        MI_TRY(A(&x_skip[i], nf)); MI_TRY(A(&x_skip_t[i], nt));
        big = std::max(big, std::max(nf, nt));
        const size_t hpad = round_up(hCh[i] / 4, 16);
        hid = std::max(hid, std::max(hpad * hFr[i + 1] * T, hpad * (size_t)lp[i + 1]));
    }
    hid = std::max(hid, big / 2);
    MI_TRY(A(&x_skip[4], C4 * T)); MI_TRY(A(&x_skip[5], C5 * T5)); MI_TRY(A(&x_inject, C4 * T));
    MI_TRY(A(&x_a, big)); MI_TRY(A(&x_b, big)); MI_TRY(A(&x_h, hid)); MI_TRY(A(&x_ta, big)); MI_TRY(A(&x_tb, big)); MI_TRY(A(&x_th, hid));
    if (c.dtype != MI_DTYPE_F32) {
        MI_TRY(A(&x_gimg, big / 2)); MI_TRY(A(&x_tgimg, big / 2));         // half modes: operand images of the decoders' GLU outputs
        for (int i = 0; i < 3; ++i) {            // ... and phase-split images of the encoder outputs 0..2 (gemm_conv.h MI_FLAG_IMG4)
            const size_t pqf = (size_t)(hFr[i + 1] / 4 + 1) * T, pqt = round_up(ceil_div(lt[i + 1], 4) + 1, 4);
            x_eimg_floats[0][i] = (2 * hCh[i] * pqf + 64) * B; x_eimg_floats[1][i] = (2 * hCh[i] * pqt + 64) * B;
            MI_TRY(A(&x_eimg[0][i], 2 * hCh[i] * pqf)); MI_TRY(A(&x_eimg[1][i], 2 * hCh[i] * pqt));
        }
    }
    MI_HIP(hipMemset(x_h, 0, (hid + 64) * B * sizeof(float)));        // hidden tensors carry zero padding channels
    MI_HIP(hipMemset(x_th, 0, (hid + 64) * B * sizeof(float)));
    const size_t zsz = std::max<size_t>(C6 * (T5 + 2), std::max<size_t>(C6 * T + 64, C3 * (4 * T + 8)));   // largest: decoder.1's 384 x 8 x T
    MI_TRY(A(&x_zA, zsz)); MI_TRY(A(&x_zB, zsz));
    MI_TRY(A(&x_a4, C5 * (T5 + 2))); MI_TRY(A(&x_b4, C5 * (T5 + 2))); MI_TRY(A(&x_pre, C5 * (T5 + 2)));
    const size_t fw = 2 * T + 600;                      // frames x 200 steps of the BLSTM chunking
    MI_TRY(A(&x_dh, C3 * T)); MI_TRY(A(&x_dy1, C3 * T)); MI_TRY(A(&x_dy2, C3 * T)); MI_TRY(A(&x_dy3, C3 * T));
    MI_TRY(A(&x_xf, C3 * fw)); MI_TRY(A(&x_gx, C6 * fw)); MI_TRY(A(&x_o0, C4 * fw)); MI_TRY(A(&x_o1, C4 * fw)); MI_TRY(A(&x_xl, C3 * fw));
    MI_TRY(A(&x_qkc, (3 * C3 + 16) * T)); MI_TRY(A(&x_att, C3 * T));
    MI_TRY(A(&x_lstm, 6 * C3 * (fw / 200 + 2)));          // LSTM state: h ping / pong and c for every (direction, sequence)
    MI_TRY(halloc(&x_lstm_scratch, lstm_persist_scratch_bytes()));
    MI_HIP(hipHostMalloc((void **)&lstm_timeout, 64, hipHostMallocMapped));
    *lstm_timeout = 0;
    // (the decoder inputs carry 128 bytes of slack in front as well: the float32 k x k convs read them by LDS-DMA in runs shifted
    // by one sample, gemm_conv.hip conv_gemm_dmatap_kernel; A() already leaves 64 floats per item behind)
    auto AS = [&](float **p, size_t per_item) { const int r = halloc((void **)p, ((per_item + 64) * B + 64) * sizeof(float)); if (r == MI_OK) *p += 32; return r; };
    MI_TRY(A(&x_dec[0], C4 * T)); MI_TRY(AS(&x_dec[1], C3 * 8 * T)); MI_TRY(AS(&x_dec[2], hCh[2] * 32 * T)); MI_TRY(AS(&x_dec[3], hCh[1] * 128 * T));
    MI_TRY(AS(&x_dec[4], hCh[0] * 512 * T)); MI_TRY(A(&x_dec[5], (size_t)4 * S * 2048 * T));
    MI_TRY(AS(&x_tdec[0], (size_t)C3 * lp[4])); MI_TRY(AS(&x_tdec[1], (size_t)hCh[2] * lp[3])); MI_TRY(AS(&x_tdec[2], (size_t)hCh[1] * lp[2]));
    MI_TRY(AS(&x_tdec[3], (size_t)hCh[0] * lp[1])); MI_TRY(A(&x_tdec[4], (size_t)2 * S * lp[0]));
    MI_TRY(A(&x_yt, (size_t)4 * S * 2048 * T)); MI_TRY(A(&x_fr, (size_t)S * T * 2 * 4096));
    const size_t max_rows = B * 512;
    x_stats_bytes = max_rows * kStatSlots * 2 * sizeof(double);
    MI_TRY(halloc((void **)&x_stats, x_stats_bytes)); MI_TRY(halloc((void **)&x_stats_t, x_stats_bytes));
    MI_HIP(hipMemset(x_stats, 0, x_stats_bytes)); MI_HIP(hipMemset(x_stats_t, 0, x_stats_bytes));
    gram2_bytes = B * ((size_t)4 << 20);                  // Model::run_dconv's Gram accumulators: B x 512 rows x 32 x 32 float64 at layer 0
    MI_TRY(halloc((void **)&w_gram2, gram2_bytes));
    MI_HIP(hipMemset(w_gram2, 0, gram2_bytes));
    gram2t_bytes = B * ((size_t)2 << 20);                 // the waveform branch's own accumulators: it runs on the side stream
    MI_TRY(halloc((void **)&w_gram2_t, gram2t_bytes));
    MI_HIP(hipMemset(w_gram2_t, 0, gram2t_bytes));
    MI_TRY(halloc((void **)&x_st1, max_rows * sizeof(float2))); MI_TRY(halloc((void **)&x_st2, max_rows * sizeof(float2)));
    MI_TRY(halloc((void **)&x_st1t, max_rows * sizeof(float2))); MI_TRY(halloc((void **)&x_st2t, max_rows * sizeof(float2)));
    MI_TRY(halloc((void **)&x_nf, B * sizeof(float2))); MI_TRY(halloc((void **)&x_df, B * sizeof(float2)));
    MI_TRY(halloc((void **)&x_nt, B * sizeof(float2))); MI_TRY(halloc((void **)&x_dt, B * sizeof(float2)));
    MI_HIP(hipDeviceSynchronize());
    return MI_OK;
}

int HModel::ktab(HGeo &g, const Gather &ga, int Kpad, const mi_ktab_entry **out) {
    char key[160];
    snprintf(key, sizeof(key), "%d %d %d %d %d %d %d %lld %d %d", ga.Cin, ga.K1, ga.K2, ga.dil1, ga.dil2, ga.pad1, ga.pad2,
             (long long)ga.chan_stride, ga.D2, Kpad);
    auto it = g.ktabs.find(key);
    if (it == g.ktabs.end()) {
        mi_ktab_entry *t = nullptr;
        const int64_t before = device_bytes;
        MI_TRY(make_ktab(ga, Kpad, &t));
        g.bytes += device_bytes - before;
        it = g.ktabs.emplace(key, t).first;
    }
    *out = it->second;
    return MI_OK;
}

// Drops the least recently used geometry: its gather tables are freed (hipFree waits for the device, so no launch that
// still reads them is in flight) and taken off the handle's allocation list.
void HModel::evict_lru() {
    auto victim = geos.end();
    for (auto it = geos.begin(); it != geos.end(); ++it)
        if (victim == geos.end() || it->second.last_use < victim->second.last_use) victim = it;
    if (victim == geos.end()) return;
    (void)hipDeviceSynchronize();
    for (auto &kv : victim->second.ktabs) {
        void *p = const_cast<mi_ktab_entry *>(kv.second);
        auto a = std::find(allocs.begin(), allocs.end(), p);
        if (a != allocs.end()) allocs.erase(a);
        (void)hipFree(p);
    }
    device_bytes -= victim->second.bytes;
    geos.erase(victim);
}

int HModel::geometry(int L, HGeo **out) {
    auto it = geos.find(L);
    if (it == geos.end()) {
        while (geos.size() >= kMaxGeos) evict_lru();
        HGeo g;
        g.L = L; g.T = (L + 1023) / 1024; g.T5 = (g.T + 1) / 2; g.Tp = std::max(32, round_up(g.T, 4));
        g.Lt[0] = L;
        for (int i = 0; i < 5; ++i) g.Lt[i + 1] = (g.Lt[i] + 3) / 4;
        for (int i = 0; i < 6; ++i) g.Lp[i] = round_up(g.Lt[i], 4);
        MI_REQUIRE(g.Lt[5] == g.T, "time branch (%d) and spectrogram (%d frames) disagree", g.Lt[5], g.T);
        HGeo &G = g;                                    // built completely before it enters the cache: a failed table
                                                        // upload leaves no half-made geometry for later forwards to hit
        for (int i = 0; i < 4; ++i) {                   // DConv of layers 0-3: this geometry's gather tables
            const int C = hCh[i], T = G.Tp;
            for (int br = 0; br < 2; ++br) {
                DConvW dw = br ? htenc[i].dconv : henc[i].dconv;
                const int64_t cs = br ? (int64_t)G.Lp[i + 1] : (int64_t)hFr[i + 1] * T;
                const int D2 = br ? G.Lp[i + 1] : T;
                for (int d = 0; d < 2; ++d) {
                    const int dil = 1 << d, hp = round_up(dw.h, 16);
                    const mi_ktab_entry *k3, *k1;
                    MI_TRY(ktab(G, Gather{C, 1, 3, 1, dil, 0, dil, cs, D2}, dw.l[d].conv3.Kpad, &k3));
                    MI_TRY(ktab(G, Gather{hp, 1, 1, 1, 1, 0, 0, cs, D2}, dw.l[d].conv1.Kpad, &k1));
                    dw.l[d].ktab3 = const_cast<mi_ktab_entry *>(k3);
                    dw.l[d].ktab1 = const_cast<mi_ktab_entry *>(k1);
                }
                (br ? G.tenc_dconv[i] : G.enc_dconv[i]) = dw;
            }
        }
        MI_HIP(hipDeviceSynchronize());
        it = geos.emplace(L, std::move(g)).first;
    }
    it->second.last_use = ++use_clock;
    *out = &it->second;
    return MI_OK;
}

// GroupNorm(G, C) of x (B, C, in_len) [contiguous rows of pitch in_pitch == in_len] + the fused apply of hkernels.hip
int HModel::group_norm(const float *x, int B, int C, int G, int in_pitch, int in_len, int off, const float *w, const float *b, int glu,
                       int gelu, const float *scale, const float *res, int res_pitch, float *y, int Cout, int out_len, int out_pitch,
                       hipStream_t st, int chan_div) {
    MI_REQUIRE(in_pitch == in_len, "group_norm: statistics need contiguous channel rows");
    const int64_t cnt = (int64_t)(C / G) * in_len;
    MI_REQUIRE(B * G <= cfg.max_batch * 512, "group_norm: too many statistic rows");
    MI_TRY(launch_row_stats(x, B * G, cnt, cnt, x_stats, st));
    MI_TRY(launch_finalize_stats(x_stats, B * G, (double)cnt, 1e-5f, 0, x_st1, nullptr, st));
    return launch_gn_apply(x, B, C, G, in_pitch, off, x_st1, w, b, glu, gelu, scale, res, res_pitch, y, Cout, out_len, out_pitch, st, chan_div);
}

int HModel::run_lstm(const float *gx, const float *whh, int N, int H, int W, float *out, hipStream_t st) {
    if (H != 192 && H != 384) return launch_lstm_small(gx, whh, N, H, W, out, st);
    if (lstm_step_chain()) return launch_lstm_seq(gx, whh, N, H, W, out, x_lstm, st);
    return launch_lstm_persist(gx, whh, N, H, W, out, x_lstm_scratch, lstm_timeout, st);
}

// DConv with BLSTM + LocalState (layers 4, 5): rows are batch items, x (B, C, Tn) contiguous; result back in x
int HModel::run_deep(HGeo &g, HEncW &e, int C, int Tn, int B, float *x, float *tmp, hipStream_t st) {
    const int H = C / 4;
    const Geo gt{B, 1, Tn, 0};
    // half modes: the k = 3 convs of layers 4 / 5 and of decoders 0 / 1 (K up to 4 608) on the tap-DMA route of gemm_tap.hip, their
    // float32 inputs converted to operand images by one streaming pass each (MI_H_NO_DEEP_TAP=1: table-driven gathers)
    static const bool no_deep_tap = getenv("MI_H_NO_DEEP_TAP") != nullptr;
    const bool deep_tap = cfg.dtype != MI_DTYPE_F32 && !no_deep_tap;
    float *src = x, *dst = tmp;
    for (int d = 0; d < 2; ++d) {
        const HDeepLayerW &l = e.deep[d];
        const int dil = 1 << d;
        const mi_ktab_entry *k;
        MI_TRY(ktab(g, Gather{C, 1, 3, 1, dil, 0, dil, (int64_t)Tn, Tn}, l.conv3.Kpad, &k));
        mi_conv_desc c3 = base_desc(l.conv3, k, src, (int64_t)C * Tn, gt);
        c3.epi = MI_EPI_LINEAR; c3.y = x_dh; c3.y_bstride = (int64_t)H * Tn; c3.y_cstride = Tn;
        if (deep_tap && l.conv3.wtap) {          // K = 3 C = 2 304 / 4 608: the taps of a 16-bit image of the layer input, by LDS-DMA
            MI_TRY(launch_f32_to_image(src, B, C, Tn, cfg.dtype, x_zB, st));
            c3.xh = x_zB; c3.xh_n = (int64_t)B * Tn; c3.wtap = l.conv3.wtap; c3.ntaps = 3; c3.tap_k2 = 3; c3.tap_pad2 = dil; c3.tap_dil2 = dil;
        }
        MI_TRY(conv(c3, st));
        MI_TRY(group_norm(x_dh, B, H, 1, Tn, Tn, 0, l.g1w, l.g1b, 0, 1, nullptr, nullptr, 0, x_dy1, H, Tn, Tn, st));
        // ---- BLSTM(hidden, layers = 2, max_steps = 200, skip) ------------------------------------------------------
        const bool framed = Tn > 200;
        const int F = framed ? (Tn + 99) / 100 : 1, W = framed ? 200 : Tn, N = B * F;
        const float *xin = x_dy1;
        if (framed) { MI_TRY(launch_unfold_frames(x_dy1, B, H, Tn, F, 200, 100, x_xf, st)); xin = x_xf; }
        const Geo gs{N, 1, W, 0};
        for (int layer = 0; layer < 2; ++layer) {
            const int Kin = layer ? 2 * H : H;
            MI_TRY(ktab(g, Gather{Kin, 1, 1, 1, 1, 0, 0, (int64_t)W, W}, l.ih[layer].Kpad, &k));
            mi_conv_desc gi = base_desc(l.ih[layer], k, layer ? x_o0 : xin, (int64_t)Kin * W, gs);
            gi.plain = 1; gi.epi = MI_EPI_LINEAR; gi.y = x_gx; gi.y_bstride = (int64_t)8 * H * W; gi.y_cstride = W;
            MI_TRY(conv(gi, st));
            if (prof.on) {         // the LSTM recurrence: W dependent launches timed as one span (bench.py's latency roofline of the mode)
                const int cls = 103;         // own row: 101 is Model::run_dconv's dconv_row_kernel
                Profiler::Pending p{cls, prof.get(), prof.get(), (double)W * 2.0 * 8.0 * H * H * N, (double)W * (8.0 * H * H + 8.0 * H * N) * 4.0, W};
                MI_HIP(hipEventRecord(p.a, st));
                MI_TRY(run_lstm(x_gx, l.whhT[layer], N, H, W, layer ? x_o1 : x_o0, st));
                MI_HIP(hipEventRecord(p.b, st));
                prof.pending.push_back(p);
                snprintf(prof.rows[cls].name, sizeof(prof.rows[cls].name), lstm_step_chain() ? "lstm_step_kernel" : "lstm_persist_kernel");
            } else
            MI_TRY(run_lstm(x_gx, l.whhT[layer], N, H, W, layer ? x_o1 : x_o0, st));
        }
        MI_TRY(ktab(g, Gather{2 * H, 1, 1, 1, 1, 0, 0, (int64_t)W, W}, l.lin.Kpad, &k));
        mi_conv_desc li = base_desc(l.lin, k, x_o1, (int64_t)2 * H * W, gs);
        li.plain = 1; li.epi = MI_EPI_LINEAR;
        if (framed) {
            li.y = x_xl; li.y_bstride = (int64_t)H * W; li.y_cstride = W;
            MI_TRY(conv(li, st));
            MI_TRY(launch_restitch_frames(x_xl, B, H, Tn, F, 200, 100, x_dy1, x_dy2, st));
        } else {
            li.flags = MI_FLAG_RES; li.res = x_dy1; li.y = x_dy2; li.y_bstride = (int64_t)H * Tn; li.y_cstride = Tn;
            MI_TRY(conv(li, st));
        }
        // ---- LocalState(hidden, heads = 4, ndecay = 4) --------------------------------------------------------------
        MI_TRY(ktab(g, Gather{H, 1, 1, 1, 1, 0, 0, (int64_t)Tn, Tn}, l.qkc.Kpad, &k));
        mi_conv_desc q = base_desc(l.qkc, k, x_dy2, (int64_t)H * Tn, gt);
        const int Tq = round_up(Tn, 4);                  // row pitch of the projection: 16-byte aligned key / content rows
        q.plain = 1; q.epi = MI_EPI_LINEAR; q.y = x_qkc; q.y_bstride = (int64_t)(3 * H + 16) * Tq; q.y_cstride = Tq;
        MI_TRY(conv(q, st));
        MI_TRY(launch_local_attn(x_qkc, B, H, Tn, Tq, x_att, Tn, st));
        mi_conv_desc pj = base_desc(l.proj, k, x_att, (int64_t)H * Tn, gt);
        pj.plain = 1; pj.epi = MI_EPI_LINEAR; pj.flags = MI_FLAG_RES; pj.res = x_dy2; pj.y = x_dy3; pj.y_bstride = (int64_t)H * Tn; pj.y_cstride = Tn;
        MI_TRY(conv(pj, st));
        // ---- 1x1 -> GroupNorm(1) -> GLU -> LayerScale -> + x ---------------------------------------------------------
        mi_conv_desc c1 = base_desc(l.conv1, k, x_dy3, (int64_t)H * Tn, gt);
        c1.plain = 1; c1.epi = MI_EPI_LINEAR; c1.y = x_zA; c1.y_bstride = (int64_t)2 * C * Tn; c1.y_cstride = Tn;
        MI_TRY(conv(c1, st));
        MI_TRY(group_norm(x_zA, B, 2 * C, 1, Tn, Tn, 0, l.g2w, l.g2b, 1, 0, l.ls, src, Tn, dst, C, Tn, Tn, st));
        std::swap(src, dst);
    }
    return MI_OK;       // two layers: the result is back in x
}

int HModel::hforward(const float *mix, float *out, int B, int L, hipStream_t st) {
    // sticky: set from the device (lstm.hip) when a hidden-state wait of an EARLIER forward ran out of time -- its output is garbage
    MI_REQUIRE(!lstm_timeout || *(volatile unsigned *)lstm_timeout == 0,
               "an earlier forward's persistent LSTM kernel timed out waiting for its hidden-state exchange (GPU oversubscribed by "
               "other persistent kernels?): results since then are invalid; MI_LSTM_STEPS=1 selects the one-launch-per-step recurrence");
    if (x_dirty) {
        MI_HIP(hipMemsetAsync(x_stats, 0, x_stats_bytes, st));
        MI_HIP(hipMemsetAsync(x_stats_t, 0, x_stats_bytes, st));
        MI_HIP(hipMemsetAsync(w_gram2, 0, gram2_bytes, st));
        MI_HIP(hipMemsetAsync(w_gram2_t, 0, gram2t_bytes, st));
        x_dirty = false;
    }
    const int r = hforward_impl(mix, out, B, L, st);
    if (r != MI_OK) {
        x_dirty = true;
        if (side_st) (void)hipStreamSynchronize(side_st);      // as Model::run_core: no side-stream kernel outlives a failed forward
    }
    return r;
}

int HModel::hforward_impl(const float *mix, float *out, int B, int L, hipStream_t st) {
    MI_REQUIRE(mix && out, "forward: null buffer");
    MI_REQUIRE(B >= 1 && B <= cfg.max_batch, "forward: batch %d outside [1, %d]", B, cfg.max_batch);
    MI_REQUIRE(L >= kMinLength && L <= Lmax, "forward: length %d outside [%d, %d] (chunks shorter than %d samples are not supported by this engine)",
               L, kMinLength, Lmax, kMinLength);
    HGeo *gp;
    MI_TRY(geometry(L, &gp));
    HGeo &g = *gp;
    const int C3 = hCh[3], C4 = hCh[4], C5 = hCh[5], C6 = hCh[6];        // layer widths: channels << level
    const int T = g.T, T5 = g.T5, Tp = g.Tp;
    const int *Lt = g.Lt, *Lp = g.Lp;
    const mi_ktab_entry *k;
    taps.clear();
    // The waveform branch of layers 0-3 (and of decoder layers 2-5) runs on the side stream beside the spectral branch, as in
    // Model::run_core_impl.  Round 3 measured this schedule SLOWER for this architecture (58.5 against 53.3 ms for the 3-minute
    // track): the forward was paced by 1 600 dependent LSTM step launches that more concurrent kernels only lengthened.  With the
    // recurrence in one persistent launch per sequence (lstm.hip) the side stream pays: 37.8-38.1 against 39.3-39.5 ms.
    // MI_H_ONE_STREAM=1 keeps everything on the caller's stream (A/B, profiling).
    static const bool h_two = getenv("MI_H_ONE_STREAM") == nullptr;
    const bool two = h_two && g_two_streams && side_streams() == MI_OK;
    hipStream_t stt = two ? side_st : st;
    auto fork = [&]() -> int {
        if (!two) return MI_OK;
        MI_HIP(hipEventRecord(ev_main, st));
        MI_HIP(hipStreamWaitEvent(side_st, ev_main, 0));
        return MI_OK;
    };
    auto join = [&]() -> int {
        if (!two) return MI_OK;
        MI_HIP(hipEventRecord(ev_side, side_st));
        MI_HIP(hipStreamWaitEvent(st, ev_side, 0));
        return MI_OK;
    };
    MI_TRY(fork());
    // ---- input statistics, normalisation, STFT (hdemucs.py:693-712) ----------------------------------------------
    MI_TRY(launch_row_stats(mix, B, (int64_t)2 * L, (int64_t)2 * L, x_stats_t, stt));
    MI_TRY(launch_finalize_stats(x_stats_t, B, 2.0 * L, 1e-5f, 1, x_nt, x_dt, stt));
    MI_TRY(launch_row_affine_pitch(mix, B, 2, L, Lp[0], x_nt, x_t0, stt));
    MI_TRY(launch_stft_frames(mix, B, L, fft, x_zt, x_stats, st));
    MI_TRY(launch_finalize_stats(x_stats, B, 4.0 * 2048 * T, 1e-5f, 1, x_nf, x_df, st));
    MI_TRY(launch_cac_transpose(x_zt, B, T, x_nf, x_0, st, Tp));
    // ---- encoder layers 0-3, both branches ------------------------------------------------------------------------
    // Half modes: a level's output feeds the next level's strided conv (k = 8, s = 4, pad 2) as a PHASE-SPLIT 16-bit image written by
    // the 1x1 + GLU epilogue beside the float32 skip tensor (gemm_conv.h MI_FLAG_IMG4); the conv is then a stride-1 two-tap conv
    // whose taps gemm_tap.hip fetches by LDS-DMA.  The image slots no epilogue writes are the conv's zero padding: they depend
    // on the geometry (length, batch), so the images are re-zeroed when it changes.
    bool encimg = cfg.dtype != MI_DTYPE_F32 && x_eimg[0][0];
    for (int i = 1; i < 4 && encimg; ++i) encimg = henc[i].conv.wtap && htenc[i].conv.wtap && hCh[i - 1] % 16 == 0;
    if (encimg && (eimg_L != L || eimg_B != B)) {
        for (int br = 0; br < 2; ++br)
            for (int i = 0; i < 3; ++i) MI_HIP(hipMemsetAsync(x_eimg[br][i], 0, x_eimg_floats[br][i] * sizeof(float), st));
        eimg_L = L; eimg_B = B;
        MI_TRY(fork());                        // the side stream starts behind the re-zeroing
    }
    const float *xf = x_0, *xt = x_t0;
    for (int i = 0; i < 4; ++i) {
        const int Cin = i ? hCh[i - 1] : 4, Cint = i ? hCh[i - 1] : 2, C = hCh[i];
        {
            const Geo gin{B, 1, Lt[i], 0, Lp[i]}, go{B, 1, Lt[i + 1], 0, Lp[i + 1]};
            const int64_t P = Lp[i + 1];
            MI_TRY(ktab(g, Gather{Cint, 1, 8, 1, 1, 0, 2, (int64_t)Lp[i], Lp[i]}, htenc[i].conv.Kpad, &k));
            mi_conv_desc d = base_desc(htenc[i].conv, k, xt, (int64_t)Cint * Lp[i], gin);
            d.O2 = Lp[i + 1]; d.o2_valid = Lt[i + 1]; d.S2 = 4; d.epi = MI_EPI_LINEAR; d.flags = MI_FLAG_GELU;
            d.y = x_ta; d.y_bstride = C * P; d.y_cstride = P;
            if (i && encimg) {                 // the previous level's output as a phase-split image: slots o2, o2 + 1 of every plane
                const int Qp = round_up(ceil_div(Lt[i], 4) + 1, 4);
                d.xh = x_eimg[1][i - 1]; d.xh_n = (int64_t)B * Qp; d.wtap = htenc[i].conv.wtap; d.ntaps = 2; d.tap_k2 = 2;
                d.D2 = Qp; d.x_ld = Qp; d.S2 = 1;
            }
            MI_TRY(conv(d, stt));
            MI_TRY(run_dconv(g.tenc_dconv[i], C, go, x_ta, x_tb, x_th, x_stats_t, x_st1t, x_st2t, stt, w_gram2_t, gram2t_bytes));
            MI_TRY(ktab(g, Gather{C, 1, 1, 1, 1, 0, 0, P, (int)P}, htenc[i].rewrite.Kpad, &k));
            mi_conv_desc r = base_desc(htenc[i].rewrite, k, x_ta, C * P, go);
            r.plain = 1; r.epi = MI_EPI_GLU; r.y = x_skip_t[i]; r.y_bstride = C * P; r.y_cstride = P;
            if (i < 3 && encimg) {
                const int64_t pq = round_up(ceil_div(Lt[i + 1], 4) + 1, 4);
                r.flags |= MI_FLAG_IMG4; r.yh = x_eimg[1][i]; r.yh_pq = pq; r.yh_n = (int64_t)B * pq;
            }
            MI_TRY(conv(r, stt));
            xt = x_skip_t[i];
            taps["tenc" + std::to_string(i)] = {x_skip_t[i], C * P};
        }
        {
            const Geo gin{B, hFr[i], T, 1, Tp}, go{B, hFr[i + 1], T, 1, Tp};
            const int64_t Pin = (int64_t)hFr[i] * Tp, P = (int64_t)hFr[i + 1] * Tp;
            MI_TRY(ktab(g, Gather{Cin, 8, 1, 1, 1, 2, 0, Pin, Tp}, henc[i].conv.Kpad, &k));
            mi_conv_desc d = base_desc(henc[i].conv, k, xf, Cin * Pin, gin);
            d.O1 = hFr[i + 1]; d.S1 = 4; d.epi = MI_EPI_LINEAR; d.flags = MI_FLAG_GELU;
            d.y = x_a; d.y_bstride = C * P; d.y_cstride = P;
            d.dma_rows = 1;                    // taps along the frequency axis only
            if (i && encimg) {                 // rows o1, o1 + 1 of every plane of the previous level's image
                const int Q = hFr[i] / 4 + 1;
                d.xh = x_eimg[0][i - 1]; d.xh_n = (int64_t)B * Q * Tp; d.wtap = henc[i].conv.wtap; d.ntaps = 2; d.tap_k2 = 1;
                d.D1 = Q; d.S1 = 1;
            }
            MI_TRY(conv(d, st));
            MI_TRY(run_dconv(g.enc_dconv[i], C, go, x_a, x_b, x_h, x_stats, x_st1, x_st2, st));
            MI_TRY(ktab(g, Gather{C, 1, 1, 1, 1, 0, 0, P, Tp}, henc[i].rewrite.Kpad, &k));
            mi_conv_desc r = base_desc(henc[i].rewrite, k, x_a, C * P, go);
            r.plain = 1; r.epi = MI_EPI_GLU; r.y = x_skip[i]; r.y_bstride = C * P; r.y_cstride = P;
            if (i == 0) { r.flags = MI_FLAG_EMB; r.emb = freq_emb; }
            if (i < 3 && encimg) {
                const int64_t pq = (int64_t)(hFr[i + 1] / 4 + 1) * Tp;
                r.flags |= MI_FLAG_IMG4 | MI_FLAG_TR_FREQ; r.yh = x_eimg[0][i]; r.yh_pq = pq; r.yh_n = (int64_t)B * pq;
            }
            MI_TRY(conv(r, st));
            xf = x_skip[i];
            taps["enc" + std::to_string(i)] = {x_skip[i], C * P};
        }
    }
    // ---- layer 4: tencoder.4 (conv only) is injected into encoder.4 (Fr 8 -> 1), GroupNorm(4), deep DConv ----------
    {
        MI_TRY(ktab(g, Gather{C3, 1, 8, 1, 1, 0, 2, (int64_t)Lp[4], Lp[4]}, htenc[4].conv.Kpad, &k));
        mi_conv_desc d = base_desc(htenc[4].conv, k, xt, (int64_t)C3 * Lp[4], Geo{B, 1, Lt[4], 0, Lp[4]});
        d.O2 = T; d.o2_valid = 0; d.S2 = 4; d.epi = MI_EPI_LINEAR; d.y = x_inject; d.y_bstride = (int64_t)C4 * T; d.y_cstride = T;
        MI_TRY(conv(d, stt));
        MI_TRY(join());                            // the injection and everything the waveform encoder wrote (its skips) are complete
        taps["tenc4"] = {x_inject, (int64_t)C4 * T};
        MI_TRY(ktab(g, Gather{C3, 8, 1, 1, 1, 0, 0, (int64_t)8 * Tp, Tp}, henc[4].conv.Kpad, &k));
        mi_conv_desc e = base_desc(henc[4].conv, k, xf, (int64_t)C3 * 8 * Tp, Geo{B, 8, T, 1, Tp});
        e.O1 = 1; e.O2 = T; e.o2_valid = 0; e.S1 = 4; e.epi = MI_EPI_LINEAR; e.flags = MI_FLAG_RES; e.res = x_inject;      // output rows: exact T
        e.y = x_zA; e.y_bstride = (int64_t)C4 * T; e.y_cstride = T;
        MI_TRY(conv(e, st));
        MI_TRY(group_norm(x_zA, B, C4, 4, T, T, 0, henc[4].n1w, henc[4].n1b, 0, 1, nullptr, nullptr, 0, x_a4, C4, T, T, st));
        MI_TRY(run_deep(g, henc[4], C4, T, B, x_a4, x_b4, st));
        MI_TRY(ktab(g, Gather{C4, 1, 1, 1, 1, 0, 0, (int64_t)T, T}, henc[4].rewrite.Kpad, &k));
        mi_conv_desc r = base_desc(henc[4].rewrite, k, x_a4, (int64_t)C4 * T, Geo{B, 1, T, 0});
        r.plain = 1; r.epi = MI_EPI_LINEAR; r.y = x_zA; r.y_bstride = (int64_t)C5 * T; r.y_cstride = T;
        MI_TRY(conv(r, st));
        MI_TRY(group_norm(x_zA, B, C5, 4, T, T, 0, henc[4].n2w, henc[4].n2b, 1, 0, nullptr, nullptr, 0, x_skip[4], C4, T, T, st));
        taps["enc4"] = {x_skip[4], (int64_t)C4 * T};
    }
    // ---- layer 5: Conv1d(768 -> 1536, k 4, s 2, p 1) on the frame axis ------------------------------------------------
    {
        MI_TRY(ktab(g, Gather{C4, 1, 4, 1, 1, 0, 1, (int64_t)T, T}, henc[5].conv.Kpad, &k));
        mi_conv_desc d = base_desc(henc[5].conv, k, x_skip[4], (int64_t)C4 * T, Geo{B, 1, T, 0});
        d.O2 = T5; d.o2_valid = 0; d.S2 = 2; d.epi = MI_EPI_LINEAR; d.y = x_zA; d.y_bstride = (int64_t)C5 * T5; d.y_cstride = T5;
        MI_TRY(conv(d, st));
        MI_TRY(group_norm(x_zA, B, C5, 4, T5, T5, 0, henc[5].n1w, henc[5].n1b, 0, 1, nullptr, nullptr, 0, x_a4, C5, T5, T5, st));
        MI_TRY(run_deep(g, henc[5], C5, T5, B, x_a4, x_b4, st));
        MI_TRY(ktab(g, Gather{C5, 1, 1, 1, 1, 0, 0, (int64_t)T5, T5}, henc[5].rewrite.Kpad, &k));
        mi_conv_desc r = base_desc(henc[5].rewrite, k, x_a4, (int64_t)C5 * T5, Geo{B, 1, T5, 0});
        r.plain = 1; r.epi = MI_EPI_LINEAR; r.y = x_zA; r.y_bstride = (int64_t)C6 * T5; r.y_cstride = T5;
        MI_TRY(conv(r, st));
        MI_TRY(group_norm(x_zA, B, C6, 4, T5, T5, 0, henc[5].n2w, henc[5].n2b, 1, 0, nullptr, nullptr, 0, x_skip[5], C5, T5, T5, st));
        taps["enc5"] = {x_skip[5], (int64_t)C5 * T5};
    }
    // ---- decoder.0: input = 0 + skip (hdemucs.py:742-747) ------------------------------------------------------------
    {
        MI_TRY(ktab(g, Gather{C5, 1, 3, 1, 1, 0, 1, (int64_t)T5, T5}, hdec[0].rewrite.Kpad, &k));
        mi_conv_desc r = base_desc(hdec[0].rewrite, k, x_skip[5], (int64_t)C5 * T5, Geo{B, 1, T5, 0});
        r.epi = MI_EPI_LINEAR; r.y = x_zA; r.y_bstride = (int64_t)C6 * T5; r.y_cstride = T5;
        static const bool no_deep_tap = getenv("MI_H_NO_DEEP_TAP") != nullptr;
        if (cfg.dtype != MI_DTYPE_F32 && !no_deep_tap && hdec[0].rewrite.wtap) {
            MI_TRY(launch_f32_to_image(x_skip[5], B, C5, T5, cfg.dtype, x_zB, st));
            r.xh = x_zB; r.xh_n = (int64_t)B * T5; r.wtap = hdec[0].rewrite.wtap; r.ntaps = 3; r.tap_k2 = 3; r.tap_pad2 = 1;
        }
        MI_TRY(conv(r, st));
        MI_TRY(group_norm(x_zA, B, C6, 4, T5, T5, 0, hdec[0].n1w, hdec[0].n1b, 1, 0, nullptr, nullptr, 0, x_pre, C5, T5, T5, st));
        const int Lu = 2 * T5 + 2;                         // un-cropped ConvTranspose1d(k 4, s 2) output: the GroupNorm sees all of it
        MI_TRY(ktab(g, Gather{C5, 1, 2, 1, -1, 0, 0, (int64_t)T5, T5}, hdec[0].convtr.Kpad, &k));
        mi_conv_desc t = base_desc(hdec[0].convtr, k, x_pre, (int64_t)C5 * T5, Geo{B, 1, T5, 0});
        t.O2 = T5 + 1; t.o2_valid = 0; t.epi = MI_EPI_CONVTR; t.tr_stride = 2; t.tr_pad = 0; t.out_len = Lu;
        t.y = x_zB; t.y_cstride = Lu; t.y_bstride = (int64_t)C4 * Lu;
        MI_TRY(conv(t, st));
        // crop [1 : 1 + T], GELU, and the next layer's `x + skip`
        MI_TRY(group_norm(x_zB, B, C4, 4, Lu, Lu, 1, hdec[0].n2w, hdec[0].n2b, 0, 1, nullptr, x_skip[4], T, x_dec[0], C4, T, T, st));
        taps["dec0+skip"] = {x_dec[0], (int64_t)C4 * T};
    }
    // ---- decoder.1 (Fr 1 -> 8) and tdecoder.0 (the "empty" layer fed with decoder.1's pre-transposed-conv tensor) --------
    {
        MI_TRY(ktab(g, Gather{C4, 1, 3, 1, 1, 0, 1, (int64_t)T, T}, hdec[1].rewrite.Kpad, &k));
        mi_conv_desc r = base_desc(hdec[1].rewrite, k, x_dec[0], (int64_t)C4 * T, Geo{B, 1, T, 0});
        r.epi = MI_EPI_LINEAR; r.y = x_zA; r.y_bstride = (int64_t)C5 * T; r.y_cstride = T;
        static const bool no_deep_tap = getenv("MI_H_NO_DEEP_TAP") != nullptr;
        if (cfg.dtype != MI_DTYPE_F32 && !no_deep_tap && hdec[1].rewrite.wtap) {
            MI_TRY(launch_f32_to_image(x_dec[0], B, C4, T, cfg.dtype, x_zB, st));
            r.xh = x_zB; r.xh_n = (int64_t)B * T; r.wtap = hdec[1].rewrite.wtap; r.ntaps = 3; r.tap_k2 = 3; r.tap_pad2 = 1;
        }
        MI_TRY(conv(r, st));
        MI_TRY(group_norm(x_zA, B, C5, 4, T, T, 0, hdec[1].n1w, hdec[1].n1b, 1, 0, nullptr, nullptr, 0, x_pre, C4, T, T, st));
        MI_TRY(ktab(g, Gather{C4, 2, 1, -1, 1, 0, 0, (int64_t)T, T}, hdec[1].convtr.Kpad, &k));
        mi_conv_desc t = base_desc(hdec[1].convtr, k, x_pre, (int64_t)C4 * T, Geo{B, 1, T, 1});
        t.O1 = 2; t.epi = MI_EPI_CONVTR; t.flags = MI_FLAG_TR_FREQ; t.tr_stride = 4; t.tr_pad = 0; t.out_len = 8;
        t.y = x_zB; t.y_cstride = (int64_t)8 * T; t.y_bstride = (int64_t)C3 * 8 * T;
        MI_TRY(conv(t, st));
        // (channel, frequency row) pairs as rows of T frames: the output and the skip carry the padded pitch Tp
        MI_TRY(group_norm(x_zB, B, C3 * 8, 4, T, T, 0, hdec[1].n2w, hdec[1].n2b, 0, 1, nullptr, x_skip[3], Tp, x_dec[1], C3 * 8, T, Tp, st, 8));
        taps["dec1+skip"] = {x_dec[1], (int64_t)C3 * 8 * Tp};
        const int Lu = 4 * T + 4;                          // un-cropped ConvTranspose1d(k 8, s 4)
        MI_TRY(ktab(g, Gather{C4, 1, 2, 1, -1, 0, 0, (int64_t)T, T}, htdec[0].convtr.Kpad, &k));
        mi_conv_desc tt = base_desc(htdec[0].convtr, k, x_pre, (int64_t)C4 * T, Geo{B, 1, T, 0});
        tt.O2 = T + 1; tt.o2_valid = 0; tt.epi = MI_EPI_CONVTR; tt.tr_stride = 4; tt.tr_pad = 0; tt.out_len = Lu;
        tt.y = x_zA; tt.y_cstride = Lu; tt.y_bstride = (int64_t)C3 * Lu;
        MI_TRY(conv(tt, st));
        MI_TRY(group_norm(x_zA, B, C3, 4, Lu, Lu, 2, htdec[0].n2w, htdec[0].n2b, 0, 1, nullptr, x_skip_t[3], Lp[4], x_tdec[0], C3, Lt[4], Lp[4], st));
        taps["tdec0+skip"] = {x_tdec[0], (int64_t)C3 * Lp[4]};
    }
    // ---- decoder.2-5 / tdecoder.1-4: rewrite 3x3 (k 3) + GLU -> ConvTranspose (+ GELU + next skip) -------------------------
    // Half modes (round 4, as model.hip's decoders): every tensor between these layers feeds NOTHING BUT the next matrix product
    // (there is no DConv here: dconv_mode = 1), so it exists only as a 16-bit operand image -- the layer input (written by the
    // previous transposed conv's GELU + skip epilogue; for the first layer converted from GroupNorm's float32 output), and the
    // GLU output (written by the rewrite conv's epilogue, MI_FLAG_IMG) -- and both convs gather their taps by LDS-DMA
    // (gemm_tap.hip) instead of walking a table over float32 tensors.  MI_NO_TAP_IMAGE=1 restores the table-driven route.
    bool tapimg = cfg.dtype != MI_DTYPE_F32 && x_gimg;
    for (int j = 2; j < 6 && tapimg; ++j)        // every level has its tap-ordered weights (widths that are multiples of 8)
        tapimg = hdec[j].rewrite.wtap && htdec[j - 1].rewrite.wtap && hdec[j].convtr.wtap && htdec[j - 1].convtr.wtap;
    // the outermost transposed conv (K = 96, bound by its output) on the image route too: no conversion pass is needed here (the GLU
    // epilogue writes the image), 38.6-38.9 against 39.3-39.5 ms; MI_H_NO_LAST_TAP=1: table-driven gather over float32
    static const bool last_tap = getenv("MI_H_NO_LAST_TAP") == nullptr;
    MI_TRY(fork());
    if (tapimg) {
        MI_TRY(launch_f32_to_image(x_dec[1], B, C3, (int64_t)8 * Tp, cfg.dtype, x_b, st));
        MI_TRY(launch_f32_to_image(x_tdec[0], B, C3, (int64_t)Lp[4], cfg.dtype, x_tb, stt));
    }
    for (int j = 2; j < 6; ++j) {
        const int i = 5 - j, C = hCh[i], Fr = hFr[i + 1];
        const bool last = j == 5;
        {
            const Geo gg{B, Fr, T, 1, Tp};
            const int64_t P = (int64_t)Fr * Tp;
            MI_TRY(ktab(g, Gather{C, 3, 3, 1, 1, 1, 1, P, Tp}, hdec[j].rewrite.Kpad, &k));
            mi_conv_desc r = base_desc(hdec[j].rewrite, k, x_dec[j - 1], C * P, gg);
            r.epi = MI_EPI_GLU; r.y = x_a; r.y_bstride = C * P; r.y_cstride = P;
            const bool tr_tap = tapimg && (!last || last_tap);           // this layer's transposed conv reads an image
            r.ntaps = 9; r.tap_k2 = 3; r.tap_pad1 = 1; r.tap_pad2 = 1;       // the conv's geometry: the DMA routes need no table
            if (tapimg) {
                r.xh = j == 2 ? (const void *)x_b : (const void *)x_dec[j - 1]; r.xh_n = (int64_t)B * P;
                r.wtap = hdec[j].rewrite.wtap;
                if (tr_tap) { r.flags |= MI_FLAG_IMG; r.yh = x_gimg; r.yh_n = (int64_t)B * P; }
            }
            MI_TRY(conv(r, st));
            const int Cout = last ? 4 * S : hCh[i - 1];
            MI_TRY(ktab(g, Gather{C, 2, 1, -1, 1, 0, 0, P, Tp}, hdec[j].convtr.Kpad, &k));
            mi_conv_desc t = base_desc(hdec[j].convtr, k, x_a, C * P, gg);
            t.O1 = Fr + 1; t.epi = MI_EPI_CONVTR; t.flags = MI_FLAG_TR_FREQ; t.out_len = 4 * Fr;
            t.y_cstride = (int64_t)4 * Fr * Tp; t.y_bstride = Cout * t.y_cstride; t.y = x_dec[j];
            t.dma_rows = 1;                    // rows q and q - 1
            if (!last) { t.flags |= MI_FLAG_GELU | MI_FLAG_RES; t.res = x_skip[i - 1]; }
            if (!last && tapimg) { t.flags |= MI_FLAG_IMG; t.yh = x_dec[j]; t.yh_n = (int64_t)B * t.y_cstride; }
            if (tr_tap) { t.xh = x_gimg; t.xh_n = (int64_t)B * P; t.wtap = hdec[j].convtr.wtap; t.ntaps = 2; t.tap_k2 = 1; t.tap_dil1 = -1; }
            MI_TRY(conv(t, st));
            taps[std::string("dec") + std::to_string(j) + (last ? "" : "+skip")] = {x_dec[j], t.y_bstride};
        }
        {
            const int Lv = Lt[i + 1], Lq = Lp[i + 1], Lout = Lt[i], Lpo = Lp[i];
            const Geo gg{B, 1, Lv, 0, Lq};
            MI_TRY(ktab(g, Gather{C, 1, 3, 1, 1, 0, 1, (int64_t)Lq, Lq}, htdec[j - 1].rewrite.Kpad, &k));
            mi_conv_desc r = base_desc(htdec[j - 1].rewrite, k, x_tdec[j - 2], (int64_t)C * Lq, gg);
            r.epi = MI_EPI_GLU; r.y = x_ta; r.y_bstride = (int64_t)C * Lq; r.y_cstride = Lq;
            const bool tr_tap = tapimg && (!last || last_tap);
            r.ntaps = 3; r.tap_k2 = 3; r.tap_pad1 = 0; r.tap_pad2 = 1;
            if (tapimg) {
                r.xh = j == 2 ? (const void *)x_tb : (const void *)x_tdec[j - 2]; r.xh_n = (int64_t)B * Lq;
                r.wtap = htdec[j - 1].rewrite.wtap;
                if (tr_tap) { r.flags |= MI_FLAG_IMG; r.yh = x_tgimg; r.yh_n = (int64_t)B * Lq; }
            }
            MI_TRY(conv(r, stt));
            const int Cout = last ? 2 * S : hCh[i - 1];
            MI_TRY(ktab(g, Gather{C, 1, 2, 1, -1, 0, 0, (int64_t)Lq, Lq}, htdec[j - 1].convtr.Kpad, &k));
            mi_conv_desc t = base_desc(htdec[j - 1].convtr, k, x_ta, (int64_t)C * Lq, gg);
            t.O2 = Lv + 1; t.o2_valid = 0; t.epi = MI_EPI_CONVTR; t.out_len = Lout;
            t.y_cstride = Lpo; t.y_bstride = (int64_t)Cout * Lpo; t.y = x_tdec[j - 1];
            if (!last) { t.flags |= MI_FLAG_GELU | MI_FLAG_RES; t.res = x_skip_t[i - 1]; }
            if (!last && tapimg) { t.flags |= MI_FLAG_IMG; t.yh = x_tdec[j - 1]; t.yh_n = (int64_t)B * t.y_cstride; }
            if (tr_tap) { t.xh = x_tgimg; t.xh_n = (int64_t)B * Lq; t.wtap = htdec[j - 1].convtr.wtap; t.ntaps = 2; t.tap_k2 = 2; t.tap_dil2 = -1; }
            MI_TRY(conv(t, stt));
            taps[std::string("tdec") + std::to_string(j - 1) + (last ? "" : "+skip")] = {x_tdec[j - 1], t.y_bstride};
        }
    }
    // ---- de-normalise, iSTFT, add the time branch (hdemucs.py:770-793) -------------------------------------------------
    MI_TRY(join());
    return launch_istft(x_dec[5], B, S, L, x_df, x_tdec[4], x_dt, fft, x_yt, x_fr, out, st, Lp[0], Tp);
}

}  // namespace mi
