// A neighbour that does nothing but issue matrix instructions for a while: which MFMA stream, running in ANOTHER process,
// corrupts this package's fp32 kernels?     ./mfma_neighbour <variant> [seconds]
//   0: v_mfma_f32_32x32x2_f32     1: v_mfma_f32_32x32x16_bf16     2: v_mfma_f32_16x16x32_bf16
//   3: v_mfma_f32_32x32x16_bf16 with the accumulators in AGPRs (inline asm)     4: like 1 but only 64 registers live (1 accumulator)
//   5: like 3, operands rewritten by VALU every iteration     6: like 3, operands re-read from LDS (ds_read_b128) every iteration
//   9: like 6 with double-buffered operand registers (the LDS data lands while the MFMAs read the OTHER set)
//  10: like 6 with 48 idle cycles between the last MFMA and the ds_reads      11: like 5 with 48 idle cycles before the VALU rewrite
//   7: like 3 plus ~70 VALU instructions per 24 MFMAs (the split arithmetic)     8: like 6 plus 7 plus ds_write_b128
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ __launch_bounds__(256, 2) void spin(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
    float fa = 0.01f * lane, fb = 0.5f;
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int j = 0; j < 4; ++j) { for (int r = 0; r < 16; ++r) acc[j][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[j][r] = 0.f; }
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0x3c003c00u + (i & 255);
    __syncthreads();
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    float w[8];
    for (int e = 0; e < 8; ++e) w[e] = 0.001f * (lane + e);
    bf16x8 a2 = a, b2 = b;
    for (int it = 0; it < iters; ++it) {
        if (V == 5) {
            u32x4 z = __builtin_bit_cast(u32x4, a);
            for (int e = 0; e < 4; ++e) z[e] = (z[e] + 0x00010001u) & 0x3f7f3f7fu;
            a = __builtin_bit_cast(bf16x8, z);
            u32x4 y = __builtin_bit_cast(u32x4, b);
            for (int e = 0; e < 4; ++e) y[e] = (y[e] ^ 0x00010001u);
            b = __builtin_bit_cast(bf16x8, y);
        }
        if (V == 10 || V == 11) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
        if (V == 11) {
            u32x4 z = __builtin_bit_cast(u32x4, a);
            for (int e = 0; e < 4; ++e) z[e] = (z[e] + 0x00010001u) & 0x3f7f3f7fu;
            a = __builtin_bit_cast(bf16x8, z);
            u32x4 y = __builtin_bit_cast(u32x4, b);
            for (int e = 0; e < 4; ++e) y[e] = (y[e] ^ 0x00010001u);
            b = __builtin_bit_cast(bf16x8, y);
        }
        if (V == 6 || V == 8 || V == 10) {
            a = *reinterpret_cast<const bf16x8 *>(&lds[((it * 64 + lane) * 4) & 4095]);
            b = *reinterpret_cast<const bf16x8 *>(&lds[((it * 64 + lane) * 4 + 1024) & 4095]);
            if (V == 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (V == 9) {                                  // a2 / b2 were loaded during the previous iteration's MFMAs
            a = a2; b = b2;
        }
        if (V == 7 || V == 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 h = (__bf16)w[e]; const float r = w[e] - (float)h; const __bf16 m = (__bf16)r; const float q2 = r - (float)m;
                w[e] = w[e] * 1.0001f + q2 + (float)m * 0.5f;
            }
        }
        if (V == 8) *reinterpret_cast<float4 *>(&lds[(threadIdx.x * 4) & 4095]) = make_float4(w[0], w[1], w[2], w[3]);
        if (V == 9) {
            a2 = *reinterpret_cast<const bf16x8 *>(&lds[(((it + 1) * 64 + lane) * 4) & 4095]);
            b2 = *reinterpret_cast<const bf16x8 *>(&lds[(((it + 1) * 64 + lane) * 4 + 1024) & 4095]);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (V == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[j], 0, 0, 0);
                if (V == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                if (V == 2) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[j], 0, 0, 0);
                if (V == 3 || V >= 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
                if (V == 4) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
            }
    }
    if (V == 3 || V >= 5) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    float s = w[0] + w[5];
    for (int j = 0; j < 4; ++j) { for (int r = 0; r < 16; ++r) s += acc[j][r]; for (int r = 0; r < 4; ++r) s += acc4[j][r]; }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int V>
static void run(double seconds) {
    float *out; (void)hipMalloc(&out, 4096);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(spin<V>, dim3(512), dim3(256), 0, 0, out, 4000);
        (void)hipDeviceSynchronize();
        launches += 8;
    }
    printf("mfma_neighbour variant %d: %ld launches\n", V, launches);
}
int main(int argc, char **argv) {
    const int v = argc > 1 ? atoi(argv[1]) : 1;
    const double s = argc > 2 ? atof(argv[2]) : 12.0;
    switch (v) { case 0: run<0>(s); break; case 1: run<1>(s); break; case 2: run<2>(s); break; case 3: run<3>(s); break; case 4: run<4>(s); break;
                 case 5: run<5>(s); break; case 6: run<6>(s); break; case 7: run<7>(s); break; case 8: run<8>(s); break;
                 case 9: run<9>(s); break; case 10: run<10>(s); break; default: run<11>(s); }
    return 0;
}
