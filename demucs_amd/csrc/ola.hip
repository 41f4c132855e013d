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

// A workgroup of the two overlap-add kernels owns kOlaSpan consecutive positions, thread i the positions i + 256 e: every access
// stays a fully coalesced 4-byte one (segment offsets are odd sample counts: no wider alignment exists on the model-output side),
// the loads of a thread's kOlaE positions are in flight together and the per-workgroup search for the overlapping segments is
// paid once per 1 024 positions (one position per thread, round 1-3: 2.0-2.3 TB/s on plain streaming traffic).
constexpr int kOlaE = 4, kOlaSpan = 256 * kOlaE;

// acc[row][p] += sum over items (ascending) of weight[p - off] * out[item][row][trim + p - off]
// grid (ceil(span/kOlaSpan), rows)
__global__ __launch_bounds__(256) void ola_accumulate_kernel(float *__restrict__ acc, int64_t acc_len, const float *__restrict__ mo,
                                                             int rows, int valid, const int64_t *__restrict__ offs,
                                                             const int32_t *__restrict__ lens, const int32_t *__restrict__ trim, int B,
                                                             int64_t span_lo, int64_t span_hi, const float *__restrict__ weight,
                                                             int weight_len) {
    // the items that overlap this workgroup's positions, in ascending item order (= the reference's summation
    // order), found once per workgroup instead of B range checks per sample
    __shared__ int n_hit;
    __shared__ int hit[256];
    const int64_t p0 = span_lo + (int64_t)blockIdx.x * kOlaSpan;
    if (threadIdx.x == 0) n_hit = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 256) {           // B <= 256 in practice: one round
        const int i = base + threadIdx.x;
        const bool over = i < B && offs[i] < p0 + kOlaSpan && offs[i] + lens[i] > p0;
        const unsigned long long m = __ballot(over);
        // wave-ordered compaction keeps ascending item order: waves append in order through the barrier sequence below
        for (int w = 0; w < 4; ++w) {
            if ((threadIdx.x >> 6) == w && over) hit[n_hit + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1))] = i;
            __syncthreads();
            if (threadIdx.x == w * 64) n_hit += __popcll(m);
            __syncthreads();
        }
    }
    const int row = blockIdx.y;
    const int64_t lim = span_hi < acc_len ? span_hi : acc_len;
    float a[kOlaE];
    bool touched[kOlaE];
#pragma unroll
    for (int e = 0; e < kOlaE; ++e) {
        const int64_t p = p0 + e * 256 + threadIdx.x;
        a[e] = p < lim ? acc[(size_t)row * acc_len + p] : 0.f;
        touched[e] = false;
    }
    const int nh = n_hit;
    for (int h = 0; h < nh; ++h) {
        const int i = hit[h];
        const int64_t off_i = offs[i];
        const int len_i = lens[i], trim_i = trim[i];
        const float *src_row = mo + ((size_t)i * rows + row) * valid;
#pragma unroll
        for (int e = 0; e < kOlaE; ++e) {
            const int64_t p = p0 + e * 256 + threadIdx.x;
            const int64_t j = p - off_i;
            // lens / trim are device arrays the host cannot validate without a sync: clamp to the buffers' extents
            const int64_t src = trim_i + j;
            if (p < lim && j >= 0 && j < len_i && j < weight_len && src >= 0 && src < valid) {
                const float v = src_row[src];
                a[e] = __fadd_rn(a[e], __fmul_rn(weight[j], v));
                touched[e] = true;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < kOlaE; ++e)
        if (touched[e]) acc[(size_t)row * acc_len + p0 + e * 256 + threadIdx.x] = a[e];
}

// acc[row][p] /= sum_weight[p], sum_weight rebuilt in ascending-offset float32 order.
// offs must be sorted ascending.  grid (ceil(acc_len/kOlaSpan), rows)
__global__ __launch_bounds__(256) void ola_finish_kernel(float *__restrict__ acc, int64_t acc_len, int64_t acc_off0,
                                                         const int64_t *__restrict__ offs, const int32_t *__restrict__ lens, int n,
                                                         int max_len, const float *__restrict__ weight) {
    // one binary search per workgroup (first segment that can still cover the workgroup's first position)
    __shared__ int lo_s;
    if (threadIdx.x == 0) {
        const int64_t pb = acc_off0 + (int64_t)blockIdx.x * kOlaSpan;
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (offs[mid] > pb - max_len) hi = mid; else lo = mid + 1; }
        lo_s = lo;
    }
    __syncthreads();
    const int lo = lo_s;                                  // segments before lo end at or before pb <= p: they add nothing
    const int64_t q0 = (int64_t)blockIdx.x * kOlaSpan + threadIdx.x;
    float v[kOlaE], sw[kOlaE];
#pragma unroll
    for (int e = 0; e < kOlaE; ++e) {
        const int64_t q = q0 + e * 256;
        v[e] = q < acc_len ? acc[(size_t)blockIdx.y * acc_len + q] : 0.f;
        sw[e] = 0.f;
    }
    const int64_t p_last = acc_off0 + (int64_t)blockIdx.x * kOlaSpan + kOlaSpan - 1;
    for (int i = lo; i < n && offs[i] <= p_last; ++i) {
        const int64_t off_i = offs[i];
        const int len_i = lens[i];
#pragma unroll
        for (int e = 0; e < kOlaE; ++e) {
            const int64_t j = acc_off0 + q0 + e * 256 - off_i;
            if (j >= 0 && j < len_i && j < max_len) sw[e] = __fadd_rn(sw[e], weight[j]);
        }
    }
#pragma unroll
    for (int e = 0; e < kOlaE; ++e) {
        const int64_t q = q0 + e * 256;
        if (q < acc_len) acc[(size_t)blockIdx.y * acc_len + q] = __fdiv_rn(v[e], sw[e]);
    }
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
                          const float *weight, int weight_len, hipStream_t st) {
    MI_REQUIRE(span_hi > span_lo && span_lo >= 0, "ola: empty span");
    MI_REQUIRE(B >= 1 && B <= 256, "ola: %d segments per call (at most 256)", B);
    hipLaunchKernelGGL(ola_accumulate_kernel, dim3(ceil_div(span_hi - span_lo, kOlaSpan), rows), dim3(256), 0, st, acc, acc_len, model_out,
                       rows, valid, offs_dev, lens_dev, trim_dev, B, span_lo, span_hi, weight, weight_len);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

int launch_ola_finish(float *acc, int64_t acc_len, int rows, int64_t acc_off0, const int64_t *offs_dev, const int32_t *lens_dev,
                      int n_segments, int max_len, const float *weight, hipStream_t st) {
    hipLaunchKernelGGL(ola_finish_kernel, dim3(ceil_div(acc_len, kOlaSpan), rows), dim3(256), 0, st, acc, acc_len, acc_off0, offs_dev,
                       lens_dev, n_segments, max_len, weight);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
