// Does global_load_lds_dwordx4 accept a source address that is only 4-byte aligned?  hipcc --offload-arch=gfx950 -O2 -o dma_unaligned dma_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
__global__ void k(const float *src, float *dst, int shift) {
    __shared__ __attribute__((aligned(16))) float s[256];
    __builtin_amdgcn_global_load_lds((gvoid_t *)(src + shift + threadIdx.x * 4), (lvoid_t *)s, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) dst[i] = s[i];
}
int main() {
    float h[512], *d, *o, r[256];
    for (int i = 0; i < 512; ++i) h[i] = (float)i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(o, 0, sizeof(r));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + shift);
        printf("shift %d floats: %s, %d mismatches (first values %g %g %g %g)\n", shift, hipGetErrorString(e), bad, r[0], r[1], r[2], r[3]);
    }
    return 0;
}
