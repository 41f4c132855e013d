// Pieces shared by the two engines (model.hip: htdemucs family; hmodel.hip: Hybrid Demucs v3): weight lookup, gather
// tables, activation-tensor geometry and the descriptor skeleton of a conv launch.
#pragma once
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "gemm_conv.h"
#include "model.h"

namespace mi {

struct WeightTable {
    std::map<std::string, std::pair<const float *, int64_t>> t;
    int get(const std::string &name, int64_t numel, const float **out) const {
        auto it = t.find(name);
        if (it == t.end()) return set_error(MI_EWEIGHT, "missing tensor '%s'", name.c_str());
        if (it->second.second != numel)
            return set_error(MI_EWEIGHT, "tensor '%s' has %lld elements, expected %lld", name.c_str(),
                             (long long)it->second.second, (long long)numel);
        *out = it->second.first;
        return MI_OK;
    }
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// geometry of one gather table
struct Gather {
    int Cin, K1, K2, dil1, dil2, pad1, pad2;
    int64_t chan_stride;
    int D2;
};

static inline std::vector<mi_ktab_entry> build_ktab(const Gather &g, int Kpad) {
    std::vector<mi_ktab_entry> tab(Kpad);
    const int K = g.Cin * g.K1 * g.K2;
    for (int k = 0; k < Kpad; ++k) {
        mi_ktab_entry e;
        if (k < K) {
            const int ci = k / (g.K1 * g.K2), r = k % (g.K1 * g.K2), k1 = r / g.K2, k2 = r % g.K2;
            e.d1 = k1 * g.dil1 - g.pad1;
            e.d2 = k2 * g.dil2 - g.pad2;
            e.off = (int32_t)(ci * g.chan_stride + (int64_t)e.d1 * g.D2 + e.d2);
            e.ci = ci;
        } else {                       // K padding: never valid (weights are zero there as well)
            e.d1 = -(1 << 29); e.d2 = -(1 << 29); e.off = 0; e.ci = 0;
        }
        tab[k] = e;
    }
    return tab;
}

struct Geo {          // geometry of one activation tensor family
    int B, D1, D2;    // batch, rows (freq bins or 1), columns (frames / samples)
    int row_mode;     // 1: DConv / GroupNorm rows are (b, d1) (frequency branch), 0: b
    int ld;           // row pitch in floats (>= D2, multiple of 4 when it differs): time-branch rows are padded
    int pitch() const { return ld ? ld : D2; }
};

static inline mi_conv_desc base_desc(const PackedConv &pc, const mi_ktab_entry *ktab, const float *x, int64_t x_bs, const Geo &g) {
    mi_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.wt = pc.wt; d.M = pc.M; d.Mpad = pc.Mpad; d.K = pc.K; d.Kpad = pc.Kpad; d.ktab = ktab; d.bias = pc.bias; d.tile_m = pc.tile; d.wx = pc.wx;
    d.wh = pc.wh; d.half = pc.wh ? pc.half : 0; d.ktab_len = round_up(pc.Kpad, 32);
    d.x = x; d.x_bstride = x_bs; d.B = g.B; d.D1 = g.D1; d.D2 = g.D2; d.O1 = g.D1; d.O2 = g.pitch(); d.S1 = 1; d.S2 = 1;
    d.o2_valid = g.pitch() != g.D2 ? g.D2 : 0;       // enumerate the padded row, mask the padding columns
    d.x_ld = g.pitch() != g.D2 ? g.pitch() : 0;        // ... and the gather / plain loaders step rows by the pitch
    d.row_mode = g.row_mode;
    return d;
}


}  // namespace mi
