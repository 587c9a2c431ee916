// Micro-benchmark: sustained rate of dense f16 MFMA shapes under the package power cap (operands in registers).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        floatx16 acc[NACC];
        for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
        for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
    } else {
        floatx4 acc[NACC];
        for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) acc[n][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
        for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) s += acc[n][e];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int NACC>
void run(const char* name, float* d, int wgs, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 20; ++l) hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(wgs), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 20.0 * wgs * 4 * (double)iters * NACC * (SHAPE == 32 ? 32768.0 : 16384.0);
        printf("%s wgs=%d: %.1f ms, %.1f TFLOP/s\n", name, wgs, ms, flops / ms / 1e9);
    }
}

int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    run<32, 4>("32x32x16 f16 (4 acc)", d, 256, 20000);
    run<16, 8>("16x16x32 f16 (8 acc)", d, 256, 20000);
    run<32, 4>("32x32x16 f16 (4 acc)", d, 512, 10000);
    run<16, 8>("16x16x32 f16 (8 acc)", d, 512, 10000);
    return 0;
}
