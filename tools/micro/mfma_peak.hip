// Achievable fp32 MFMA rate on the box: a loop of independent v_mfma_f32_32x32x2_f32 with no memory traffic,
// at 1..3 waves per SIMD, with and without LDS fragment reads in the loop.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSREAD>
__global__ __launch_bounds__(256) void peak(float *out, int iters, int ldsbytes_unused) {
    __shared__ float sm[16 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16 * 256; i += 256) sm[i] = 1e-3f * (i & 15);
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a0 = 1e-3f * lane, a1 = 2e-3f, b0 = 1e-3f, b1 = 3e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (LDSREAD) {
                a0 = sm[(2 * s + (lane >> 5)) * 256 + (lane & 31)];
                a1 = sm[(2 * s + (lane >> 5)) * 256 + 32 + (lane & 31)];
                b0 = sm[(2 * s + (lane >> 5)) * 256 + 128 + (lane & 31)];
                b1 = sm[(2 * s + (lane >> 5)) * 256 + 160 + (lane & 31)];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[tid] = s;
}

template <int LDSREAD>
static void run(int blocks_per_cu, int dyn_lds) {
    float *out;
    hipMalloc(&out, 4096);
    const int iters = 4096, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(peak<LDSREAD>, dim3(grid), dim3(256), dyn_lds, 0, out, iters, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(peak<LDSREAD>, dim3(grid), dim3(256), dyn_lds, 0, out, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double flops = (double)grid * 4 * iters * 32 * 4096.0;
    printf("lds_reads=%d blocks/CU=%d (waves/SIMD=%d): %.3f ms  %.1f TFLOP/s\n", LDSREAD, blocks_per_cu, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    // occupancy is limited through dynamic LDS: 16 KiB static + dyn; 160 KiB per CU
    run<0>(1, 0); run<0>(2, 0); run<0>(3, 0); run<0>(4, 0);
    run<1>(1, 0); run<1>(2, 0); run<1>(3, 0); run<1>(4, 0);
    return 0;
}
