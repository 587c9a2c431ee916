// Faithful replica of one pass-A / pass-B unit stream in asm: 6 MFMAs per unit on two alternating accumulators, B fragments loaded by
// ds_read_b128 two units ahead (3 rotating buffers), s_waitcnt lgkmcnt before their first use, NV v_add fillers per unit.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define UNIT(BH, BL, NH, NL, F0, F1, F2, F3, F4, F5) \
    "s_waitcnt lgkmcnt(4)\n v_mfma_f32_16x16x32_f16 %[c0], %[w0], %[" BH "], %[c0]\n ds_read_b128 %[" NH "], %[addr]\n" F0 \
    "v_mfma_f32_16x16x32_f16 %[c1], %[w2], %[" BH "], %[c1]\n ds_read_b128 %[" NL "], %[addr] offset:128\n" F1 \
    "s_waitcnt lgkmcnt(4)\n v_mfma_f32_16x16x32_f16 %[c0], %[w0], %[" BL "], %[c0]\n" F2 \
    "v_mfma_f32_16x16x32_f16 %[c1], %[w2], %[" BL "], %[c1]\n" F3 \
    "v_mfma_f32_16x16x32_f16 %[c0], %[w1], %[" BH "], %[c0]\n" F4 \
    "v_mfma_f32_16x16x32_f16 %[c1], %[w3], %[" BH "], %[c1]\n" F5
#define A1 "v_add_f32 %[f0], %[f0], %[f0]\n"
#define A2 "v_add_f32 %[f0], %[f0], %[f0]\n v_add_f32 %[f1], %[f1], %[f1]\n"
#define E ""
template <int NV>
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ w, float* out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    half8 w0, w1, w2, w3;
    { const half8* h = reinterpret_cast<const half8*>(w) + lane; w0 = h[0]; w1 = h[64]; w2 = h[128]; w3 = h[192]; }
    floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    half8 bA0 = w0, bA1 = w1, bB0 = w2, bB1 = w3, bC0 = w0, bC1 = w1;
    float f0 = 1.f, f1 = 0.5f;
    const unsigned addr = ((threadIdx.x & 15) * 288 + (lane >> 4) * 16) & 0xffff;
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<uint4*>(lds)[i] = w[i & 1023];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (NV == 0)
            asm volatile(UNIT("bA0", "bA1", "bC0", "bC1", E, E, E, E, E, E) UNIT("bB0", "bB1", "bA0", "bA1", E, E, E, E, E, E) UNIT("bC0", "bC1", "bB0", "bB1", E, E, E, E, E, E)
                         : [c0] "+a"(c0), [c1] "+a"(c1), [f0] "+v"(f0), [f1] "+v"(f1), [bA0] "+v"(bA0), [bA1] "+v"(bA1), [bB0] "+v"(bB0), [bB1] "+v"(bB1), [bC0] "+v"(bC0), [bC1] "+v"(bC1)
                         : [w0] "a"(w0), [w1] "a"(w1), [w2] "a"(w2), [w3] "a"(w3), [addr] "v"(addr) : "memory");
        else if (NV == 4)
            asm volatile(UNIT("bA0", "bA1", "bC0", "bC1", E, E, A1, A1, A1, A1) UNIT("bB0", "bB1", "bA0", "bA1", E, E, A1, A1, A1, A1) UNIT("bC0", "bC1", "bB0", "bB1", E, E, A1, A1, A1, A1)
                         : [c0] "+a"(c0), [c1] "+a"(c1), [f0] "+v"(f0), [f1] "+v"(f1), [bA0] "+v"(bA0), [bA1] "+v"(bA1), [bB0] "+v"(bB0), [bB1] "+v"(bB1), [bC0] "+v"(bC0), [bC1] "+v"(bC1)
                         : [w0] "a"(w0), [w1] "a"(w1), [w2] "a"(w2), [w3] "a"(w3), [addr] "v"(addr) : "memory");
        else if (NV == 6)
            asm volatile(UNIT("bA0", "bA1", "bC0", "bC1", A1, A1, A1, A1, A1, A1) UNIT("bB0", "bB1", "bA0", "bA1", A1, A1, A1, A1, A1, A1) UNIT("bC0", "bC1", "bB0", "bB1", A1, A1, A1, A1, A1, A1)
                         : [c0] "+a"(c0), [c1] "+a"(c1), [f0] "+v"(f0), [f1] "+v"(f1), [bA0] "+v"(bA0), [bA1] "+v"(bA1), [bB0] "+v"(bB0), [bB1] "+v"(bB1), [bC0] "+v"(bC0), [bC1] "+v"(bC1)
                         : [w0] "a"(w0), [w1] "a"(w1), [w2] "a"(w2), [w3] "a"(w3), [addr] "v"(addr) : "memory");
        else
            asm volatile(UNIT("bA0", "bA1", "bC0", "bC1", A1, A1, A2, A2, A2, A2) UNIT("bB0", "bB1", "bA0", "bA1", A1, A1, A2, A2, A2, A2) UNIT("bC0", "bC1", "bB0", "bB1", A1, A1, A2, A2, A2, A2)
                         : [c0] "+a"(c0), [c1] "+a"(c1), [f0] "+v"(f0), [f1] "+v"(f1), [bA0] "+v"(bA0), [bA1] "+v"(bA1), [bB0] "+v"(bB0), [bB1] "+v"(bB1), [bC0] "+v"(bC0), [bC1] "+v"(bC1)
                         : [w0] "a"(w0), [w1] "a"(w1), [w2] "a"(w2), [w3] "a"(w3), [addr] "v"(addr) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = f0 + f1 + c0[0] + c1[3] + (float)bA0[0] + (float)bB1[2] + (float)bC0[1];
}
template <int NV> void run(const uint4* w, float* o, unsigned long long* clk) {
    const int iters = 20000; unsigned long long h[256];
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256), 131072, 0, w, o, iters, clk); hipDeviceSynchronize(); hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"variant\": \"unit stream\", \"valu_per_unit\": %d, \"cycles_per_mfma\": %.2f, \"cycles_per_unit\": %.1f, \"err\": %d}\n", NV, (double)h[100] / ((double)iters * 18), (double)h[100] / ((double)iters * 3), (int)hipGetLastError());
    fflush(stdout);
}
int main() {
    uint4* w; float* o; unsigned long long* clk;
    hipMalloc(&w, 1 << 20); hipMalloc(&o, 256 * 256 * 4); hipMalloc(&clk, 256 * 8);
    { unsigned short* h = (unsigned short*)malloc(1 << 20); unsigned st = 12345u; for (int i = 0; i < (1 << 19); ++i) { st = st * 1664525u + 1013904223u; unsigned r = st >> 8; h[i] = (unsigned short)(((r & 1) << 15) | ((11 + ((r >> 1) & 3)) << 10) | ((r >> 3) & 0x3ff)); } hipMemcpy(w, h, 1 << 20, hipMemcpyHostToDevice); }
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    run<0>(w, o, clk); run<4>(w, o, clk); run<6>(w, o, clk); run<10>(w, o, clk);
    return 0;
}
