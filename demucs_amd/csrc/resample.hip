// Fractional sinc resampler: `julius.resample_frac` as called by demucs.audio.convert_audio (demucs/audio.py:169-172),
// the step in front of Separator.separate_tensor when the input sample rate differs from the model's (demucs/api.py:265-266).
//
// julius (>= 0.2.3, not vendored in the reference) evaluates, for old_sr/new_sr reduced by their gcd,
//     y[n * new_sr + i] = sum_k kernel[i][k] * xpad[n * old_sr + k],   k < 2 * width + old_sr,
// on the input replicate-padded by (width, width + old_sr), i.e. a conv1d with stride old_sr and new_sr output
// channels interleaved in time.  The windowed-sinc table is built on the host (demucs_amd/audio.py) with the same float32
// operations; this kernel is the strided convolution.  HBM bound: 4 B in / 4 B out per sample; the table (≤ 372 KB for the common rate
// pairs) is served from LDS when it is small and from L2 otherwise.
#include "common.h"
#include "kernels.h"

namespace mi {

// one workgroup = 256 consecutive output samples of one row (channel); the (phase, tap) table sits in LDS
__global__ __launch_bounds__(256) void resample_frac_kernel(const float *__restrict__ x, int64_t L, const float *__restrict__ table,
                                                            int old_sr, int new_sr, int width, float *__restrict__ y, int64_t Lout,
                                                            int in_lds) {
    extern __shared__ float tab[];                       // [new_sr][klen] when it fits; else the table is read through L2
    const int klen = 2 * width + old_sr;
    if (in_lds) {
        for (int i = threadIdx.x; i < new_sr * klen; i += blockDim.x) tab[i] = table[i];
        __syncthreads();
    }
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= Lout) return;
    const float *xr = x + (size_t)blockIdx.y * L;
    const int64_t n = o / new_sr;
    const int i = (int)(o - n * new_sr);
    const int64_t base = n * old_sr - width;             // xpad[j] = x[clamp(j - width)]
    const float *kr = (in_lds ? tab : table) + (size_t)i * klen;
    float acc = 0.f;
    for (int k = 0; k < klen; ++k) {
        int64_t j = base + k;
        j = j < 0 ? 0 : (j >= L ? L - 1 : j);
        acc = fmaf(kr[k], xr[j], acc);
    }
    y[(size_t)blockIdx.y * Lout + o] = acc;
}

int launch_resample_frac(const float *x, int rows, int64_t L, const float *table, int old_sr, int new_sr, int width, float *y,
                         int64_t Lout, hipStream_t st) {
    size_t lds = (size_t)new_sr * (2 * width + old_sr) * sizeof(float);
    const int in_lds = lds <= 64 * 1024;                 // 48 kHz <-> 44.1 kHz: 127 KiB -> L2; 2:1 ratios: a few KiB -> LDS
    if (!in_lds) lds = 0;
    MI_REQUIRE(rows > 0 && rows < 65536 && L > 0 && Lout > 0, "resample: bad shape");
    hipLaunchKernelGGL(resample_frac_kernel, dim3((unsigned)((Lout + 255) / 256), rows), dim3(256), lds, st, x, L, table, old_sr,
                       new_sr, width, y, Lout, in_lds);
    MI_CHECK_LAUNCH();
    return MI_OK;
}

}  // namespace mi
