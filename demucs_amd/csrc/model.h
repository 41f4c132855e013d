// Engine state: packed weights, gather tables, workspace.  One Model per (weights, device).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "gemm_conv.h"
#include "kernels.h"

namespace mi {

extern int g_two_streams;      // model.hip: 1 = waveform branch on a side stream (default), 0 = everything on the caller's stream
struct WeightTable;
struct Gather;
struct Geo;

struct PackedConv {
    float *wt = nullptr, *bias = nullptr;
    void *wx = nullptr;            // split-bf16 tile image of wt (gemm_x6.hip) when the tile has that main loop
    void *wh = nullptr;            // bf16 / fp16 operand image of wt (gemm_half.hip) in the reduced-precision modes
    void *wtap = nullptr;          // ... and, for k x k convs fed by an operand image, the tap-ordered image (gemm_tap.hip)
    int ntaps = 0;
    int M = 0, Mpad = 0, K = 0, Kpad = 0, tile = 0, half = 0;
};

struct DConvLayerW {
    PackedConv conv3, conv1;
    mi_ktab_entry *ktab3 = nullptr, *ktab1 = nullptr;
    float *gn1_w = nullptr, *gn1_b = nullptr, *gn2_w = nullptr, *gn2_b = nullptr, *ls = nullptr;
    // implicit-GEMM route: weights that turn the Gram accumulators of the hidden activations into the second GroupNorm's
    // statistics (norms.hip: gram_finalize_kernel)
    double *gram_wt = nullptr, *gram_ct = nullptr, sum_b = 0.0, sum_bsq = 0.0;
};
struct DConvW {
    DConvLayerW l[2];
    int h = 0;                     // hidden channels: C / 8 (htdemucs, dconv_comp 8) or C / 4 (hdemucs, dconv_comp 4)
    bool has_row = false;          // frequency-branch C = 48 / 96: fused LDS-resident row kernel
    DConvRowLayer row[2];
    bool has_time = false;         // time-branch C = 48 / 96: fused three-pass VALU kernels (dconv_time.hip)
    DConvTimeLayer tl[2];
};

struct EncW {
    PackedConv conv, rewrite;
    mi_ktab_entry *ktab_conv = nullptr, *ktab_rw = nullptr;
    DConvW dconv;
};
struct DecW {
    PackedConv rewrite, convtr;
    mi_ktab_entry *ktab_rw = nullptr, *ktab_tr = nullptr;
    DConvW dconv;
};
struct TrLayerW {
    PackedConv qkv_proj, q_proj, kv_proj, out_proj, lin1, lin2;
    // LayerNorm folded into the projection that consumes it: c1[m] = sum_k (W diag(ln_w))[m][k] (see MI_FLAG_LN)
    float *qkv_c1 = nullptr, *q_c1 = nullptr, *kv_c1 = nullptr, *lin1_c1 = nullptr;
    float *norm_w[4] = {}, *norm_b[4] = {};   // norm1, norm2, norm3 (cross only), norm_out
    float *gamma1 = nullptr, *gamma2 = nullptr;
};

// Per-kernel-class device timing with HIP events on the launch stream (bench.py's roofline leg).
struct ProfRow {
    int cls = 0;
    char name[48] = "";
    long long launches = 0;
    double ms = 0.0, flops = 0.0, bytes = 0.0;
};
struct Profiler {
    bool on = false;
    struct Pending { int cls; hipEvent_t a, b; double flops, bytes; long long launches = 1; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    ProfRow rows[128];
    hipEvent_t get();
    void begin();
    int end(hipStream_t st);
    ~Profiler();
};

// Activation workspace of one forward (sized for max_batch segments).  Handles created on the same device with the
// same geometry SHARE one workspace (a bag of four fine-tuned models holds the ~0.57 GB per batched segment once,
// not four times): such handles must not run concurrently -- they are used from one stream, one after the other,
// which is what the bag loop of apply_model does.
struct WorkspacePtrs {
    float *w_xt0 = nullptr, *w_zt = nullptr, *w_x0 = nullptr;
    float *w_skip[4] = {}, *w_skip_t[4] = {};
    void *w_eimg[2][3] = {};     // half modes: phase-split operand images of the encoder outputs 0..2 ([branch][level], gemm_conv.h MI_FLAG_IMG4)
    float *w_a = nullptr, *w_b = nullptr, *w_c = nullptr, *w_h = nullptr;
    float *w_ta = nullptr, *w_tb = nullptr, *w_tc = nullptr, *w_th = nullptr;
    float2 *w_tr_stat[2][2] = {}, *w_tr_stat1[2] = {};   // per-token (mean, rstd) of the layer inputs / of x1
    float *w_tr_x[2][2] = {}, *w_tr_qkv[2] = {}, *w_tr_att[2] = {}, *w_tr_x1[2] = {},
          *w_tr_x2[2] = {}, *w_tr_ffh[2] = {};
    // half modes: 16-bit operand images [512 / 8][B tokens][8] of the layer inputs (beside w_tr_x) and of x1: what the
    // in-projections and lin1 move global -> LDS by DMA
    float *w_tr_ximg[2][2] = {}, *w_tr_x1img[2] = {};
    float *w_yspec = nullptr, *w_ytime = nullptr, *w_yt = nullptr, *w_fr = nullptr;
    double *w_stats = nullptr, *w_stats_t = nullptr, *w_gram = nullptr, *w_gram2 = nullptr, *w_gram2_t = nullptr;
    size_t gram2t_bytes = 0;
    size_t gram2_bytes = 0;      // w_gram2: Gram accumulators of the implicit-GEMM DConv route (rows x slots x HP x HP float64)
    float2 *w_st1 = nullptr, *w_st2 = nullptr, *w_st1_t = nullptr, *w_st2_t = nullptr;
    float2 *w_norm_f = nullptr, *w_denorm_f = nullptr, *w_norm_t = nullptr, *w_denorm_t = nullptr;
};
struct Workspace : WorkspacePtrs {
    std::string key;
    std::vector<void *> allocs;
    int64_t bytes = 0;
    size_t stats_bytes = 0, gram_bytes = 0;
    // a forward that failed half-way may leave the self-cleaning statistics slots (norms.hip) non-zero: the next
    // forward re-zeroes them first instead of silently mis-normalising every later call
    bool dirty = false;
    int alloc(void **p, size_t n);
    ~Workspace();
};

struct Model : WorkspacePtrs {
    mi_config cfg{};
    Profiler prof;
    int conv(const mi_conv_desc &d, hipStream_t st);
    int attn(const float *q, const float *k, const float *v, float *o, int B, int Tq, int Tk, int64_t q_bs, int64_t kv_bs,
             int64_t o_bs, hipStream_t st, bool image = false);
    int attn_heads(const void *q, const void *k, const void *v, float *o, int B, int Tq, int Tk, hipStream_t st);
    int S = 0, SL = 0, T = 0, Lt[5] = {};   // time-branch lengths per level
    int Lp[5] = {};                         // ... and their row pitches (rounded up to 4 floats: 16-byte aligned rows)
    int64_t device_bytes = 0;
    std::vector<void *> allocs;

    FftTables fft{};
    EncW enc[4], tenc[4];
    DecW dec[4], tdec[4];
    float *freq_emb = nullptr;
    PackedConv chan[4];
    mi_ktab_entry *chan_ktab[4] = {};
    mi_ktab_entry *tr_ktab512[2] = {}, *tr_ktab2048[2] = {};
    float *norm_in_w[2] = {}, *norm_in_b[2] = {}, *pos_emb[2] = {};
    TrLayerW tr[2][5];

    hipStream_t side_st = nullptr;          // the waveform branch's stream (run_core_impl)
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    int side_streams();
    std::shared_ptr<Workspace> ws;   // activation workspace, shared by every handle with the same (device, geometry, max_batch)

    ~Model();
    int init(const mi_config &c, const mi_tensor_desc *weights, size_t n);
    int forward(const float *mix, float *out, int B, hipStream_t st);
    int forward_core(const float *mix, const float *mag, float *spec_out, float *time_out, int B, hipStream_t st);
    int run_core(const float *mix, const float *mag, int B, hipStream_t st);
    int run_core_impl(const float *mix, const float *mag, int B, hipStream_t st);

   protected:
    int dev_alloc(void **p, size_t bytes);
    template <typename T> int upload(const std::vector<T> &h, T **dptr);
    int pack_split(PackedConv *pc);
    int pack_half(PackedConv *pc);
    int pack_conv(const float *W, const float *bias, int M, int K, bool glu, PackedConv *pc, int ntaps = 0);
    int pack_tap(PackedConv *pc, int ntaps);
    int pack_enc_tap(const float *W, int Cin, PackedConv *pc);
    int pack_convtr(const float *W, const float *bias, int Cin, int Cout, PackedConv *pc, int stride = 4);
    int pack_vec(const float *v, int n, int npad, bool glu, float **out);
    int pack_linear_ln(const float *W, const float *bias, const float *ln_w, const float *ln_b, int M, int K, PackedConv *pc,
                       float **c1);
    int make_ktab(const Gather &g, int Kpad, mi_ktab_entry **out);
    int load_dconv(const WeightTable &wt, const std::string &prefix, int C, int64_t chan_stride, int D2, bool freq, DConvW *dw, int comp = 8);
    int alloc_workspace();
    int fill_workspace(Workspace &w);
    bool dconv_tap_dma = false;  // run_dconv may state the k = 3 convs' geometry (DMA tap route): only when x / tmp carry 128 bytes of slack
    int run_dconv(const DConvW &w, int C, const Geo &g, float *x, float *tmp, float *hidden, double *stats, float2 *st1, float2 *st2,
                  hipStream_t st, double *gram2 = nullptr, size_t gram2_cap = 0);
    int run_tr_layer(int br, int k, int B, const float *x, const float2 *xstat, const float *other, const float2 *ostat, float *out,
                     float2 *outstat, hipStream_t st, const void *ximg = nullptr, const void *oimg = nullptr, void *outimg = nullptr);
};

}  // namespace mi
