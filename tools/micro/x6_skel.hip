// Bisects the split-bf16 GEMM main loop (gemm_x6.hip, 128x128 tile): each feature of the K step can be switched
// off to see what the 24-MFMA body loses to it.  F bits: 1 = A LDS-DMA, 2 = B global loads, 4 = B split + ds_write,
// 8 = fragment ds_reads (else constant registers), 16 = barrier per step, 32 = epilogue stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

__device__ __forceinline__ void split3(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    const bf16x2 hh = {(__bf16)x0, (__bf16)x1};
    const float r0 = x0 - (float)hh[0], r1 = x1 - (float)hh[1];
    const bf16x2 mm = {(__bf16)r0, (__bf16)r1};
    const float q0 = r0 - (float)mm[0], q1 = r1 - (float)mm[1];
    const bf16x2 ll = {(__bf16)q0, (__bf16)q1};
    h = __builtin_bit_cast(unsigned, hh); m = __builtin_bit_cast(unsigned, mm); l = __builtin_bit_cast(unsigned, ll);
}

template <int F, int OCC>
__global__ __launch_bounds__(256, OCC) void skel(const unsigned char *wx, const float *x, float *out, int nk, int P, int MT) {
    constexpr int BM = 128, BN = 128, A_BYTES = BM * 96, B_BYTES = BN * 96, STAGE = A_BYTES + B_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int mt = blockIdx.x % MT, nt = blockIdx.x / MT;
    for (int i = tid; i < 2 * STAGE / 4; i += 256) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u + (i & 0xff);
    __syncthreads();
    const unsigned char *aimg = wx + (size_t)mt * nk * A_BYTES + lane * 16;
    const int bn = tid & 127, bh = __builtin_amdgcn_readfirstlane(tid >> 7);
    const float *bp = x + (size_t)nt * BN + bn + (size_t)(8 * bh) * P;
    float breg[8];
    for (int j = 0; j < 8; ++j) breg[j] = 0.001f * (tid + j);
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    bf16x8 af[2][3], bf[2][3];
    for (int p = 0; p < 3; ++p) for (int a = 0; a < 2; ++a) for (int e = 0; e < 8; ++e) { af[a][p][e] = (__bf16)(0.01f * (lane + e + p)); bf[a][p][e] = (__bf16)(0.02f * (lane - e + a)); }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char *As = smem + cur * STAGE, *Bs = As + A_BYTES;
        if (F & 8) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a][p] = *reinterpret_cast<const bf16x8 *>(As + (((p * 2 + lh) * BM) + (wm * 2 + a) * 32 + li) * 16);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b][p] = *reinterpret_cast<const bf16x8 *>(Bs + (((p * 2 + lh) * BN) + (wn * 2 + b) * 32 + li) * 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            if (F & 1) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    __builtin_amdgcn_global_load_lds((gvoid_t *)(aimg + (size_t)(kt + 1) * A_BYTES + (c * 4 + wave) * 1024),
                                                     (lvoid_t *)(smem + (cur ^ 1) * STAGE + (c * 4 + wave) * 1024), 16, 0, 0);
            }
            if (F & 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) breg[j] = bp[(size_t)(kt + 1) * 16 * P + (size_t)j * P];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA[q]], bf[b][PB[q]], acc[a][b], 0, 0, 0);
        if ((F & 4) && kt + 1 < nk) {
            unsigned ph[4], pm[4], pl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(breg[2 * j], breg[2 * j + 1], ph[j], pm[j], pl[j]);
            unsigned char *bs = smem + (cur ^ 1) * STAGE + A_BYTES + (bh * BN + bn) * 16;
            *reinterpret_cast<uint4 *>(bs) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            *reinterpret_cast<uint4 *>(bs + 2 * BN * 16) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
            *reinterpret_cast<uint4 *>(bs + 4 * BN * 16) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
        if (F & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (F & 16) __syncthreads();
        cur ^= 1;
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += breg[j];
    if (F & 32) {
        float *o = out + (size_t)(mt * 128 + wm * 64 + lh * 4) * P + (size_t)nt * BN + wn * 64 + li;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r)
            o[(size_t)(a * 32 + (r & 3) + 8 * (r >> 2)) * P + b * 32] = acc[a][b][r] + s;
    } else {
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
        if (s == 12345.678f) out[tid] = s;
    }
}

__global__ void fill_random(float *p, size_t n, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = ((float)(z >> 40) / 8388608.0f - 1.0f) * scale;
    }
}

static unsigned char *g_wx; static float *g_x, *g_out;
constexpr int M = 2048, K = 512, NTILES = 651, P = NTILES * 128, MT = M / 128, NK = K / 16;

template <int F, int OCC>
static void run(const char *what) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = MT * NTILES;
    hipLaunchKernelGGL((skel<F, OCC>), dim3(grid), dim3(256), 0, 0, g_wx, g_x, g_out, NK, P, MT);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((skel<F, OCC>), dim3(grid), dim3(256), 0, 0, g_wx, g_x, g_out, NK, P, MT);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = 2.0 * M * K * (double)P;
    printf("F=%2d occ=%d %-44s %8.3f ms %7.1f TF(fp32-equiv) %6.1f%% of bf16 peak\n", F, OCC, what, ms, flops / ms / 1e9, 6 * flops / ms / 1e9 / 2500 * 100);
}

int main() {
    (void)hipMalloc(&g_wx, (size_t)M * K * 6); (void)hipMalloc(&g_x, (size_t)K * P * 4); (void)hipMalloc(&g_out, (size_t)M * P * 4);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *)g_wx, (size_t)M * K * 6 / 4, 1.0f);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, g_x, (size_t)K * P, 1.7f);
    run<0, 2>("MFMA only");
    run<8, 2>("+frag reads");
    run<8 + 16, 2>("+frag reads +barrier");
    run<8 + 16 + 4, 2>("+frags +barrier +split/store");
    run<8 + 16 + 4 + 2, 2>("+frags +barrier +split/store +B loads");
    run<8 + 16 + 1, 2>("+frags +barrier +A DMA");
    run<8 + 16 + 1 + 2, 2>("+frags +barrier +A DMA +B loads (no split)");
    run<31, 2>("full main loop");
    run<63, 2>("full + epilogue stores");
    run<63, 3>("full + epilogue stores");
    run<31 - 16, 2>("full w/o barrier (racy)");
    return 0;
}
