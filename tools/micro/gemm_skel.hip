// Bisects what the 128x128 fp32 GEMM tile loses against the bare MFMA rate: the same 32-MFMA K-tile body with
// optional (BAR) one s_barrier per K tile, (DMA) 4 global_load_lds per wave per K tile from an L2-resident
// buffer, (EPI) 64 global stores per lane at the end, for grids of short (nk = 32) and long blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <int BAR, int DMA, int EPI>
__global__ __launch_bounds__(256, 3) void skel(const float *src, float *out, int nk, size_t span) {
    constexpr int SS = 16 * 256;
    __shared__ __attribute__((aligned(16))) float smem[3 * SS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 3 * SS; i += 256) smem[i] = 1e-3f * (i & 15);
    __syncthreads();
    const float *g = src + ((size_t)blockIdx.x * 4096) % span + (size_t)(4 * wave + lh) * 128 + li * 4;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#define TILE(kt, stage)                                                                                  \
    do {                                                                                                 \
        float *sa = smem + (stage) * SS + (4 * wave) * 128, *sb = sa + 16 * 128;                         \
        const float *ga = g + (size_t)(kt) * 2048, *gb = ga + 1024 * 1024;                               \
        __builtin_amdgcn_global_load_lds((gvoid_t *)ga, (lvoid_t *)sa, 16, 0, 0);                        \
        __builtin_amdgcn_global_load_lds((gvoid_t *)(ga + 256), (lvoid_t *)(sa + 256), 16, 0, 0);        \
        __builtin_amdgcn_global_load_lds((gvoid_t *)gb, (lvoid_t *)sb, 16, 0, 0);                        \
        __builtin_amdgcn_global_load_lds((gvoid_t *)(gb + 256), (lvoid_t *)(sb + 256), 16, 0, 0);        \
    } while (0)
    if (DMA) { TILE(0, 0); TILE(1, 1); }
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (DMA) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (BAR) __builtin_amdgcn_s_barrier();
        if (DMA && kt + 2 < nk) TILE(kt + 2, stage == 0 ? 2 : stage - 1);
        const float *As = smem + stage * SS, *Bs = As + 16 * 128;
        float af[2][2], bf[2][2];
        for (int a = 0; a < 2; ++a) af[0][a] = As[lh * 128 + (wm * 2 + a) * 32 + li];
        for (int b = 0; b < 2; ++b) bf[0][b] = Bs[lh * 128 + (wn * 2 + b) * 32 + li];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) {
                for (int a = 0; a < 2; ++a) af[(s + 1) & 1][a] = As[(2 * (s + 1) + lh) * 128 + (wm * 2 + a) * 32 + li];
                for (int b = 0; b < 2; ++b) bf[(s + 1) & 1][b] = Bs[(2 * (s + 1) + lh) * 128 + (wn * 2 + b) * 32 + li];
            }
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][a], bf[s & 1][b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    if (EPI) {
        float *o = out + ((size_t)blockIdx.x * 16384) % span + (size_t)(wm * 64 + lh * 4) * 128 + wn * 64 + li;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r)
            o[(size_t)(a * 32 + (r & 3) + 8 * (r >> 2)) * 128 + b * 32] = acc[a][b][r];
    } else {
        float s = 0.f;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
        if (s == 12345.678f) out[tid] = s;
    }
}

__global__ void fill_random(float *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = ((float)(z >> 40) / 8388608.0f - 1.0f) * 1.7f;
    }
}
static float *g_src, *g_out;
static const size_t SPAN = 64u << 20;   // floats

template <int BAR, int DMA, int EPI>
static void run(int grid, int nk) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((skel<BAR, DMA, EPI>), dim3(grid), dim3(256), 0, 0, g_src, g_out, nk, SPAN - (8u << 20));
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((skel<BAR, DMA, EPI>), dim3(grid), dim3(256), 0, 0, g_src, g_out, nk, SPAN - (8u << 20));
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double flops = (double)grid * 4 * nk * 32 * 4096.0;
    printf("bar=%d dma=%d epi=%d grid=%6d nk=%4d: %8.3f ms %7.1f TFLOP/s\n", BAR, DMA, EPI, grid, nk, ms, flops / ms / 1e9);
}

int main() {
    (void)hipMalloc(&g_src, SPAN * 4); (void)hipMalloc(&g_out, SPAN * 4);
    (void)hipMemset(g_src, 0, SPAN * 4);
    const int G = 10416;
    printf("-- zero inputs\n");
    run<0, 0, 0>(G, 32); run<1, 1, 0>(G, 32); run<1, 1, 1>(G, 32);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, g_src, SPAN);
    printf("-- random inputs (DMA variants read them; others compute on the LDS fill pattern)\n");
    run<0, 0, 0>(G, 32); run<1, 0, 0>(G, 32); run<0, 1, 0>(G, 32); run<1, 1, 0>(G, 32); run<1, 1, 1>(G, 32); run<0, 0, 1>(G, 32);
    run<0, 0, 0>(768, 434); run<1, 0, 0>(768, 434); run<1, 1, 0>(768, 434); run<1, 1, 1>(768, 434);
    run<1, 1, 1>(2604, 128); run<1, 1, 1>(2604, 32);
    return 0;
}
