// Fills the LDS of every CU with a bit pattern (default: bf16/fp32 NaNs) so that a kernel launched next that consumes
// LDS it never wrote shows it in its results.  hipcc --offload-arch=gfx950 -shared -fPIC -o liblds_poison.so
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pattern, unsigned *sink) {
    extern __shared__ unsigned lds[];
    const int n = 64 * 1024 / 4;                      // 64 KiB per workgroup, 2 workgroups per CU cover 128 KiB
    for (int i = threadIdx.x; i < n; i += 256) lds[i] = pattern;
    __syncthreads();
    if (pattern == 1u) sink[threadIdx.x] = lds[(threadIdx.x * 61) % n];
    // linger so that the whole grid is co-resident and every CU's LDS is touched
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
extern "C" int lds_poison(unsigned pattern, void *stream) {
    static unsigned *sink = nullptr;
    if (!sink && hipMalloc((void **)&sink, 1024) != hipSuccess) return 1;
    hipLaunchKernelGGL(lds_poison_kernel, dim3(512), dim3(256), 64 * 1024, (hipStream_t)stream, pattern, sink);
    return hipGetLastError() != hipSuccess;
}
