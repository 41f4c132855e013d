// Hybrid Demucs v3 engine state (`hdemucs_mmi` architecture: the reference's HDemucs with its default hyper-parameters,
// demucs/hdemucs.py:366-580).  Reuses the packing helpers and the conv launcher of the htdemucs engine (Model).
#pragma once
#include <map>
#include <string>

#include "model.h"
#include "model_internal.h"

namespace mi {

struct HDeepLayerW {            // one DConv layer with BLSTM + LocalState (encoder layers 4, 5; demucs/demucs.py:133-149)
    PackedConv conv3, ih[2], lin, qkc, proj, conv1;
    float *whhT[2] = {};        // per LSTM layer: W_hh of both directions in the step kernel's operand order (pack_lstm_whh)
    float *g1w = nullptr, *g1b = nullptr, *g2w = nullptr, *g2b = nullptr, *ls = nullptr;
};
struct HEncW {
    PackedConv conv, rewrite;
    float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;    // GroupNorm(4) affine (layers >= 4)
    DConvW dconv;               // layers 0..3: implicit-GEMM route, hidden = C / 4 (gather tables are per geometry)
    HDeepLayerW deep[2];        // layers 4, 5
};
struct HDecW {
    PackedConv rewrite, convtr;
    float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;
};

struct HGeo {                   // everything that depends on the input length
    int L = 0, T = 0, Tp = 0, T5 = 0, Lt[6] = {}, Lp[6] = {};    // Tp: row pitch of the frequency-branch tensors (>= 32, multiple of 4)
    std::map<std::string, const mi_ktab_entry *> ktabs;
    DConvW enc_dconv[4], tenc_dconv[4];      // copies of the weights with this geometry's gather tables
    uint64_t last_use = 0;                   // forward counter at the last use: the least recently used geometry is evicted
    int64_t bytes = 0;                       // device bytes of `ktabs`
};

struct HModel : Model {
    int Lmax = 0, Tmax = 0;
    int hCh[7] = {};             // layer widths: channels << level (48 ... 3 072 for hdemucs_mmi; 4 ... 256 for demucs_unittest)
    HEncW henc[6], htenc[5];
    HDecW hdec[6], htdec[5];
    std::map<int, HGeo> geos;              // at most kMaxGeos cached input lengths (LRU): every track's tail chunk and every
    uint64_t use_clock = 0;                // random shift brings a new length, and a long-lived handle sees many tracks
    std::vector<void *> hws;     // workspace allocations (freed with the handle)
    int64_t hws_bytes = 0;
    // workspace
    float *x_t0 = nullptr, *x_zt = nullptr, *x_0 = nullptr, *x_skip[6] = {}, *x_skip_t[4] = {}, *x_inject = nullptr;
    float *x_a = nullptr, *x_b = nullptr, *x_h = nullptr, *x_ta = nullptr, *x_tb = nullptr, *x_th = nullptr;
    float *x_zA = nullptr, *x_zB = nullptr, *x_a4 = nullptr, *x_b4 = nullptr, *x_pre = nullptr;
    float *x_dh = nullptr, *x_dy1 = nullptr, *x_dy2 = nullptr, *x_dy3 = nullptr, *x_xf = nullptr, *x_gx = nullptr, *x_o0 = nullptr,
          *x_o1 = nullptr, *x_xl = nullptr, *x_qkc = nullptr, *x_att = nullptr, *x_lstm = nullptr;
    float *x_dec[6] = {}, *x_tdec[5] = {}, *x_yt = nullptr, *x_fr = nullptr;
    float *x_eimg[2][3] = {};         // half modes: phase-split operand images of the encoder outputs 0..2 ([branch][level]); slots the
    size_t x_eimg_floats[2][3] = {};  // epilogues never write are the convs' zero padding: re-zeroed when the geometry changes
    int eimg_L = -1, eimg_B = -1;     // geometry the images' zero slots are valid for
    float *x_gimg = nullptr, *x_tgimg = nullptr;      // half modes: 16-bit operand images of the decoders' GLU outputs (hmodel.hip)
    double *x_stats = nullptr, *x_stats_t = nullptr;
    float2 *x_st1 = nullptr, *x_st2 = nullptr, *x_st1t = nullptr, *x_st2t = nullptr, *x_nf = nullptr, *x_df = nullptr, *x_nt = nullptr,
           *x_dt = nullptr;
    size_t x_stats_bytes = 0;
    bool x_dirty = false;
    void *x_lstm_scratch = nullptr;        // lstm.hip: granule buffers + control words of the persistent recurrence kernel
    unsigned *lstm_timeout = nullptr;      // pinned host word the kernel sets when a wait times out (sticky; checked by every forward)
    std::map<std::string, std::pair<const float *, int64_t>> taps;    // name -> (buffer, floats per item) of the last forward

    ~HModel();
    int hinit(const mi_config &c, const mi_tensor_desc *weights, size_t n);
    int hforward(const float *mix, float *out, int B, int L, hipStream_t st);

   private:
    int halloc(void **p, size_t bytes);
    int geometry(int L, HGeo **out);
    void evict_lru();
    int ktab(HGeo &g, const Gather &ga, int Kpad, const mi_ktab_entry **out);
    int load_deep(const WeightTable &wt, const std::string &prefix, int C, HDeepLayerW *l, int d);
    int load_norm(const WeightTable &wt, const std::string &name, int C, float **w, float **b);
    int run_lstm(const float *gx, const float *whh, int N, int H, int W, float *out, hipStream_t st);
    int run_deep(HGeo &g, HEncW &e, int C, int Tn, int B, float *x, float *tmp, hipStream_t st);
    int group_norm(const float *x, int B, int C, int G, int in_pitch, int in_len, int off, const float *w, const float *b, int glu, int gelu,
                   const float *scale, const float *res, int res_pitch, float *y, int Cout, int out_len, int out_pitch, hipStream_t st,
                   int chan_div = 1);
    int hforward_impl(const float *mix, float *out, int B, int L, hipStream_t st);
};

}  // namespace mi
