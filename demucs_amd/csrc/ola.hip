// Device side of the segment scheduler (reference: demucs/apply.py:108-124,257-301).
//
// The whole track stays resident in HBM: segments are cut out of it with TensorChunk.padded
// semantics, and the weighted overlap-add runs on the device in the reference's summation
// order (ascending segment offset, float32, product and sum rounded separately), so the stitched
// result is bit-identical to `out[..., off:off+SL] += weight[:n] * chunk_out; out /= sum_weight`
// applied to the same per-segment outputs.
#include "common.h"
#include "kernels.h"

namespace mi {

// seg[b][c][i] = track[c][starts[b] + i] or 0 outside [0, track_len).  grid (ceil(valid/256), channels, B)
__global__ __launch_bounds__(256) void segments_gather_kernel(const float *__restrict__ track, int64_t track_len, int channels,
                                                              const int64_t *__restrict__ starts, int valid,
                                                              float *__restrict__ seg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= valid) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t p = starts[b] + i;
    seg[((size_t)b * channels + c) * valid + i] = (p >= 0 && p < track_len) ? track[(size_t)c * track_len + p] : 0.f;
}

// acc[row][p] += sum over items (ascending) of weight[p - off] * out[item][row][trim + p - off]
// grid (ceil(span/256), rows)
__global__ __launch_bounds__(256) void ola_accumulate_kernel(float *__restrict__ acc, int64_t acc_len, const float *__restrict__ mo,
                                                             int rows, int valid, const int64_t *__restrict__ offs,
                                                             const int32_t *__restrict__ lens, const int32_t *__restrict__ trim, int B,
                                                             int64_t span_lo, int64_t span_hi, const float *__restrict__ weight) {
    const int64_t p = span_lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= span_hi || p >= acc_len) return;
    const int row = blockIdx.y;
    float a = acc[(size_t)row * acc_len + p];
    bool touched = false;
    for (int i = 0; i < B; ++i) {
        const int64_t j = p - offs[i];
        if (j >= 0 && j < lens[i]) {
            const float v = mo[((size_t)i * rows + row) * valid + trim[i] + j];
            a = __fadd_rn(a, __fmul_rn(weight[j], v));
            touched = true;
        }
    }
    if (touched) acc[(size_t)row * acc_len + p] = a;
}

// acc[row][p] /= sum_weight[p], sum_weight rebuilt in ascending-offset float32 order.
// offs must be sorted ascending.  grid (ceil(acc_len/256), rows)
__global__ __launch_bounds__(256) void ola_finish_kernel(float *__restrict__ acc, int64_t acc_len, int64_t acc_off0,
                                                         const int64_t *__restrict__ offs, const int32_t *__restrict__ lens, int n,
                                                         int max_len, const float *__restrict__ weight) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= acc_len) return;
    const int64_t p = acc_off0 + q;
    // first segment with off > p - max_len
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (offs[mid] > p - max_len) hi = mid; else lo = mid + 1; }
    float sw = 0.f;
    for (int i = lo; i < n && offs[i] <= p; ++i) {
        const int64_t j = p - offs[i];
        if (j < lens[i]) sw = __fadd_rn(sw, weight[j]);
    }
    const size_t idx = (size_t)blockIdx.y * acc_len + q;
    acc[idx] = __fdiv_rn(acc[idx], sw);
}

int launch_segments_gather(const float *track, int64_t track_len, int channels, const int64_t *starts_dev, int B, int valid,
                           float *seg, hipStream_t st) {
    hipLaunchKernelGGL(segments_gather_kernel, dim3(ceil_div(valid, 256), channels, B), dim3(256), 0, st, track, track_len, channels,
                       starts_dev, valid, seg);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_ola_accumulate(float *acc, int64_t acc_len, int rows, const float *model_out, int valid, const int64_t *offs_dev,
                          const int32_t *lens_dev, const int32_t *trim_dev, int B, int64_t span_lo, int64_t span_hi,
                          const float *weight, hipStream_t st) {
    MI_REQUIRE(span_hi > span_lo && span_lo >= 0, "ola: empty span");
    hipLaunchKernelGGL(ola_accumulate_kernel, dim3(ceil_div(span_hi - span_lo, 256), rows), dim3(256), 0, st, acc, acc_len, model_out,
                       rows, valid, offs_dev, lens_dev, trim_dev, B, span_lo, span_hi, weight);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_ola_finish(float *acc, int64_t acc_len, int rows, int64_t acc_off0, const int64_t *offs_dev, const int32_t *lens_dev,
                      int n_segments, int max_len, const float *weight, hipStream_t st) {
    hipLaunchKernelGGL(ola_finish_kernel, dim3(ceil_div(acc_len, 256), rows), dim3(256), 0, st, acc, acc_len, acc_off0, offs_dev,
                       lens_dev, n_segments, max_len, weight);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
