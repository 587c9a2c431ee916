// Micro-benchmark: the body sweep's chunk loop in isolation, on the two dense f16 MFMA shapes, under the package power cap.
//
// One wave per SIMD (256-thread work-group, one per CU, as body_sweep_kernel), wave tile 32 output channels x 96 rows,
// split-fp16 x3 (hi*hi + hi*lo + lo*hi).  Per chunk of 32 input channels BOTH variants issue the same operand traffic:
// 12 ds_read_b128 of activation fragments (conflict-free LDS image, random fp16 data) and 4 global_load_dwordx4 of
// weight fragments (L2-resident, random), and the same 294,912 MACs per wave:
//   SHAPE 32: 18 x v_mfma_f32_32x32x16_f16   (3 row tiles of 32, 2 k-steps of 16)          32 cycles each
//   SHAPE 16: 36 x v_mfma_f32_16x16x32_f16   (2 channel tiles x 6 row tiles of 16, k = 32)  16 cycles each
// MI355X_MICROARCH.md (DVFS give-back, item 7) reports the 16x16x32 shape at 1.12-1.15x the FLOP/s of 32x32x16 at
// equal cycles on random data; this bench checks it for THIS loop's operand mix before the sweep is restructured.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape_lds.hip -o /tmp/mfma_shape_lds && /tmp/mfma_shape_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half8 as_h8(uint4 v) { union { uint4 u; half8 h; } c; c.u = v; return c.h; }

constexpr int LDS_BYTES = 128 * 1024;
constexpr int NCHUNK = 14;              // chunks per "layer" (7 taps x 2 halves)

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void loop_kernel(const uint4* __restrict__ wfrag, const uint4* __restrict__ fill, float* out, int layers,
                                                       unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LDS_BYTES / 16; i += 256) lds[i] = fill[i];
    __syncthreads();
    const uint4* wb = wfrag + lane;
    float s = 0.f;
    // in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles / (100 MHz ticks) around the loop
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto lidx = [&](int l, int c, int i) { return ((c * 12 + i) * 64 + lane + (tid >> 6) * 7 + l * 320) & (LDS_BYTES / 16 - 1); };
    uint4 w[2][4];
    for (int f = 0; f < 4; ++f) { w[0][f] = wb[f * 64]; w[1][f] = wb[(4 + f) * 64]; }
    uint4 b0[12], b1[12];                   // activation fragments of the current / next chunk (software pipeline)
#pragma unroll
    for (int i = 0; i < 12; ++i) b0[i] = lds[lidx(0, 0, i)];
    if constexpr (SHAPE == 32) {
        floatx16 acc[3];
        for (int k = 0; k < 3; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
        auto chunk = [&](uint4 (&wc)[4], uint4 (&bc)[12], uint4 (&bn)[12], int l, int c) {
#pragma unroll
            for (int i = 0; i < 12; ++i) bn[i] = lds[lidx(l, c + 1, i)];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(wc[2 * ks]), as_h8(bc[k * 4 + 2 * ks]), acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(wc[2 * ks]), as_h8(bc[k * 4 + 2 * ks + 1]), acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(wc[2 * ks + 1]), as_h8(bc[k * 4 + 2 * ks]), acc[k], 0, 0, 0);
                }
                wc[2 * ks] = wb[(((l * NCHUNK + c + 2) & 63) * 4 + 2 * ks) * 64];
                wc[2 * ks + 1] = wb[(((l * NCHUNK + c + 2) & 63) * 4 + 2 * ks + 1) * 64];
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        for (int l = 0; l < layers; ++l) {
#pragma unroll
            for (int c = 0; c < NCHUNK; c += 2) {
                chunk(w[0], b0, b1, l, c);
                chunk(w[1], b1, b0, l, c + 1);
            }
        }
        for (int k = 0; k < 3; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
    } else {
        floatx4 acc[2][6];
        for (int m = 0; m < 2; ++m) for (int k = 0; k < 6; ++k) for (int e = 0; e < 4; ++e) acc[m][k][e] = 0.f;
        auto chunk = [&](uint4 (&wc)[4], uint4 (&bc)[12], uint4 (&bn)[12], int l, int c) {
#pragma unroll
            for (int i = 0; i < 12; ++i) bn[i] = lds[lidx(l, c + 1, i)];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wc[2 * m]), as_h8(bc[2 * k]), acc[m][k], 0, 0, 0);
                    acc[m][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wc[2 * m]), as_h8(bc[2 * k + 1]), acc[m][k], 0, 0, 0);
                    acc[m][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wc[2 * m + 1]), as_h8(bc[2 * k]), acc[m][k], 0, 0, 0);
                }
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) wc[f] = wb[(((l * NCHUNK + c + 2) & 63) * 4 + f) * 64];
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        for (int l = 0; l < layers; ++l) {
#pragma unroll
            for (int c = 0; c < NCHUNK; c += 2) {
                chunk(w[0], b0, b1, l, c);
                chunk(w[1], b1, b0, l, c + 1);
            }
        }
        for (int m = 0; m < 2; ++m) for (int k = 0; k < 6; ++k) for (int e = 0; e < 4; ++e) s += acc[m][k][e];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }    // a buffer of their own: no output depends on them
    out[blockIdx.x * 256 + tid] = s;
}

static unsigned short rnd_half(unsigned& st) {            // random fp16 in about [-1, 1): sign, exponent 11..14, random mantissa
    st = st * 1664525u + 1013904223u;
    const unsigned r = st >> 8;
    return (unsigned short)(((r & 1) << 15) | ((11 + ((r >> 1) & 3)) << 10) | ((r >> 3) & 0x3ff));
}

template <int SHAPE>
double run(const char* name, const uint4* w, const uint4* fill, float* out, int layers, int launches, unsigned long long* clk) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((loop_kernel<SHAPE>), dim3(256), dim3(256), LDS_BYTES, 0, w, fill, out, layers, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double)launches * 256 * 4 * (double)layers * NCHUNK * 294912.0;
    const double tf = 2.0 * macs / ms / 1e9;
    unsigned long long h[512];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc[256], ghz[256];
    for (int i = 0; i < 256; ++i) { cyc[i] = (double)h[2 * i] / ((double)layers * NCHUNK); ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; }
    auto med = [](double* v) { for (int i = 0; i < 256; ++i) for (int j = i + 1; j < 256; ++j) if (v[j] < v[i]) { double t = v[i]; v[i] = v[j]; v[j] = t; } return v[128]; };
    printf("{\"shape\": \"%s\", \"ms\": %.1f, \"tflops\": %.1f, \"shader_cycles_per_chunk_median\": %.1f, \"mfma_issue_cycles_per_chunk\": 576, "
           "\"in_kernel_clock_ghz_median\": %.3f}\n", name, ms, tf, med(cyc), med(ghz));
    fflush(stdout);
    return tf;
}

int main() {
    const size_t wn = 64 * 4 * 64, fn = LDS_BYTES / 16;
    uint4 *hw = (uint4*)malloc(wn * 16), *hf = (uint4*)malloc(fn * 16);
    unsigned st = 12345u;
    for (size_t i = 0; i < wn * 8; ++i) ((unsigned short*)hw)[i] = rnd_half(st);
    for (size_t i = 0; i < fn * 8; ++i) ((unsigned short*)hf)[i] = rnd_half(st);
    uint4 *dw, *df; float* out;
    hipMalloc(&dw, wn * 16); hipMalloc(&df, fn * 16); hipMalloc(&out, 256 * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 512 * 8);
    hipMemcpy(dw, hw, wn * 16, hipMemcpyHostToDevice); hipMemcpy(df, hf, fn * 16, hipMemcpyHostToDevice);
    const int layers = 4000, launches = 40;          // >= 2 s per run: the clock settles under the power cap
    for (int rep = 0; rep < 3; ++rep) {
        run<32>("32x32x16", dw, df, out, layers, launches, clk);
        run<16>("16x16x32", dw, df, out, layers, launches, clk);
    }
    return 0;
}
