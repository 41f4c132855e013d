// Do workgroups with >= 64 KiB of LDS keep their LDS to themselves when OTHER kernels share the CUs?
// Two streams: "big" workgroups (LDS size under test) and "small" ones (8 KiB) each fill their LDS with a private
// pattern, linger, and re-check it.  Any mismatch = another workgroup wrote into this one's allocation.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void holder(int words, int rounds, unsigned salt, unsigned *errors) {
    extern __shared__ unsigned lds[];
    const unsigned key = (blockIdx.x * 2654435761u) ^ salt;
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = key + i;
    __syncthreads();
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int s = 0; s < 20; ++s) __builtin_amdgcn_s_sleep(20);
        for (int i = threadIdx.x; i < words; i += 256) bad += lds[i] != key + i;
        // keep writing the TOP of the allocation, like a kernel that uses all of it
        for (int i = words - 1024 + threadIdx.x; i < words; i += 256) lds[i] = key + i;
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
}
int main() {
    unsigned *err; (void)hipMalloc(&err, 8); 
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int sizes[] = {48 * 1024, 60 * 1024, 64 * 1024 - 512, 64 * 1024, 65 * 1024, 80 * 1024};
    for (int big : sizes) {
        (void)hipFuncSetAttribute((const void *)holder, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipMemset(err, 0, 8);
        for (int rep = 0; rep < 40; ++rep) {
            hipLaunchKernelGGL(holder, dim3(512), dim3(256), big, s1, big / 4, 6, 0x1111u, err);
            hipLaunchKernelGGL(holder, dim3(2048), dim3(256), 8 * 1024, s2, 2048, 6, 0x2222u, err + 1);
            hipLaunchKernelGGL(holder, dim3(1024), dim3(256), 24 * 1024, s2, 6144, 3, 0x3333u, err + 1);
        }
        (void)hipDeviceSynchronize();
        unsigned h[2]; (void)hipMemcpy(h, err, 8, hipMemcpyDeviceToHost);
        printf("big LDS %6d B: mismatching words seen by big workgroups %u, by small ones %u  (%s)\n", big, h[0], h[1], hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
