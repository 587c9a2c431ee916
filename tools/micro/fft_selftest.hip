// Device-vs-host self test of fft_small.h: every butterfly Bf<R> and every complex primitive runs once on the GPU
// (packed-fp32 inline asm) and once on the host (the scalar formulas of the same header); the two must agree to
// rounding.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stofnet_amd/csrc tools/micro/fft_selftest.hip -o tools/micro/fft_selftest.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "fft_small.h"
#include "ct_twiddles.h"
using namespace stof_fft;

template <int R>
__global__ void bf_kernel(const float2* in, float2* out) {
    const int t = threadIdx.x;
    cf x[R];
    for (int k = 0; k < R; ++k) x[k] = mk(in[t * R + k].x, in[t * R + k].y);
    Bf<R>::run(x);
    for (int k = 0; k < R; ++k) out[t * R + k] = make_float2(x[k].x, x[k].y);
}

__global__ void prim_kernel(const float2* in, float2* out) {
    const int t = threadIdx.x;
    const cf a = mk(in[2 * t].x, in[2 * t].y), b = mk(in[2 * t + 1].x, in[2 * t + 1].y);
    cf r[8];
    r[0] = cmul(a, b); r[1] = cmulc(a, b); r[2] = add_i(a, b); r[3] = sub_i(a, b);
    r[4] = cmul_uniform(a, mk(0.5f, -0.8660254037844386f)); r[5] = fma_real(0.3f, b, a); r[6] = cscale(a, 1.5f); r[7] = csub(a, b);
    for (int k = 0; k < 8; ++k) out[8 * t + k] = make_float2(r[k].x, r[k].y);
}

static double check(const char* name, const std::vector<float2>& dev, const std::vector<float2>& ref) {
    double worst = 0;
    for (size_t i = 0; i < dev.size(); ++i) {
        const double d = fmax(fabs((double)dev[i].x - ref[i].x), fabs((double)dev[i].y - ref[i].y));
        if (!(d <= worst)) worst = d;                       // NaN propagates
    }
    printf("%-12s max |device - host| = %.3g %s\n", name, worst, worst < 1e-5 ? "ok" : "MISMATCH");
    return worst;
}

template <int R>
static double run_bf() {
    const int T = 64;
    std::vector<float2> h(T * R), d(T * R), ref(T * R);
    for (int i = 0; i < T * R; ++i) h[i] = make_float2(sinf(0.37f * i + 1.f), cosf(0.11f * i * i + 2.f));
    float2 *din, *dout;
    hipMalloc(&din, sizeof(float2) * T * R); hipMalloc(&dout, sizeof(float2) * T * R);
    hipMemcpy(din, h.data(), sizeof(float2) * T * R, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(bf_kernel<R>, dim3(1), dim3(T), 0, 0, din, dout);
    hipMemcpy(d.data(), dout, sizeof(float2) * T * R, hipMemcpyDeviceToHost);
    for (int t = 0; t < T; ++t) {                          // double-precision DFT as the reference
        for (int q = 0; q < R; ++q) {
            double re = 0, im = 0;
            for (int k = 0; k < R; ++k) {
                const double a = -2.0 * M_PI * q * k / R, c = cos(a), s = sin(a);
                re += h[t * R + k].x * c - h[t * R + k].y * s;
                im += h[t * R + k].x * s + h[t * R + k].y * c;
            }
            ref[t * R + q] = make_float2((float)re, (float)im);
        }
    }
    hipFree(din); hipFree(dout);
    char name[32]; snprintf(name, sizeof name, "Bf<%d>", R);
    return check(name, d, ref);
}

// whole transform: analytic_ct<N, 64> by one wave on a padded LDS slot vs the same passes run thread by thread on the host
template <int N>
__global__ void analytic_kernel(const float2* in, float2* out, const float2* table) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    constexpr CtPlan P = ct_plan_for(N);
    cf* const W = reinterpret_cast<cf*>(lds);
    cf* const Z = W + (P.table + 1) / 2 * 2;
    const int tid = threadIdx.x;
    if (table) { for (int i = tid; i < P.table; i += 64) lds[i] = table[i]; }
    else stof_ct::stage_twiddles<N>(lds, tid, 64);                 // the device-resident compile-time table
    for (int i = tid; i < N; i += 64) Z[pad((unsigned)i)] = mk(in[i].x, in[i].y);
    __syncthreads();
    analytic_ct<N, 64>(Z, W, tid, [] {});
    for (int i = tid; i < N; i += 64) out[i] = make_float2(Z[pad((unsigned)i)].x, Z[pad((unsigned)i)].y);
}

template <int N, int T, int S, int M>
static void host_level(cf* Z, const cf* W) {
    constexpr CtPlan P = ct_plan_for(N);
    if constexpr (S < P.npass) {
        constexpr int R = P.radix[S];
        for (int tid = 0; tid < T; ++tid) ct_pass<N, M, R, false, T>(Z, W, tid);
        host_level<N, T, S + 1, M / R>(Z, W);
        for (int tid = 0; tid < T; ++tid) ct_pass<N, M, R, true, T>(Z, W, tid);
    } else {
        for (int tid = 0; tid < T; ++tid) ct_middle16<N, T>(Z, tid, ct_filter_default<N>());
    }
}

template <int N>
static double run_analytic() {
    constexpr CtPlan P = ct_plan_for(N);
    static constexpr TwTable<P.table> table = make_tw_table<N, P.table>();
    std::vector<float2> h(N), d(N), ref(N);
    for (int i = 0; i < N; ++i) h[i] = make_float2(sinf(0.37f * i + 1.f) * expf(-1e-3f * i), cosf(0.011f * i * i + 2.f));
    std::vector<cf> slot(ct_slot_entries(N));
    for (int i = 0; i < N; ++i) slot[ct_padded(i)] = mk(h[i].x, h[i].y);
    host_level<N, 64, 0, N>(slot.data(), reinterpret_cast<const cf*>(table.w));
    for (int i = 0; i < N; ++i) ref[i] = make_float2(slot[ct_padded(i)].x, slot[ct_padded(i)].y);
    float2 *din, *dout, *dtab;
    hipMalloc(&din, sizeof(float2) * N); hipMalloc(&dout, sizeof(float2) * N); hipMalloc(&dtab, sizeof(float2) * P.table);
    hipMemcpy(din, h.data(), sizeof(float2) * N, hipMemcpyHostToDevice);
    hipMemcpy(dtab, table.w, sizeof(float2) * P.table, hipMemcpyHostToDevice);
    const size_t lds = ((P.table + 1) / 2 * 2 + ct_slot_entries(N)) * sizeof(float2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&analytic_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(analytic_kernel<N>, dim3(1), dim3(64), lds, 0, din, dout, dtab);
    hipMemcpy(d.data(), dout, sizeof(float2) * N, hipMemcpyDeviceToHost);
    char name[48]; snprintf(name, sizeof name, "ct<%d>", N);
    double w = check(name, d, ref);
    hipLaunchKernelGGL(analytic_kernel<N>, dim3(1), dim3(64), lds, 0, din, dout, (const float2*)nullptr);
    hipMemcpy(d.data(), dout, sizeof(float2) * N, hipMemcpyDeviceToHost);
    snprintf(name, sizeof name, "ct<%d> devtab", N);
    w = fmax(w, check(name, d, ref));
    hipFree(din); hipFree(dout); hipFree(dtab);
    return w;
}

int main() {
    double worst = 0;
    worst = fmax(worst, run_analytic<96>()); worst = fmax(worst, run_analytic<1536>()); worst = fmax(worst, run_analytic<2000>());
    worst = fmax(worst, run_analytic<2048>()); worst = fmax(worst, run_analytic<4000>()); worst = fmax(worst, run_analytic<4096>());
    worst = fmax(worst, run_bf<2>()); worst = fmax(worst, run_bf<3>()); worst = fmax(worst, run_bf<4>());
    worst = fmax(worst, run_bf<5>()); worst = fmax(worst, run_bf<6>()); worst = fmax(worst, run_bf<8>());
    worst = fmax(worst, run_bf<10>()); worst = fmax(worst, run_bf<16>());
    const int T = 64;
    std::vector<float2> h(2 * T), d(8 * T), ref(8 * T);
    for (int i = 0; i < 2 * T; ++i) h[i] = make_float2(sinf(0.7f * i + 1.f), cosf(0.3f * i + 2.f));
    float2 *din, *dout;
    hipMalloc(&din, sizeof(float2) * 2 * T); hipMalloc(&dout, sizeof(float2) * 8 * T);
    hipMemcpy(din, h.data(), sizeof(float2) * 2 * T, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(prim_kernel, dim3(1), dim3(T), 0, 0, din, dout);
    hipMemcpy(d.data(), dout, sizeof(float2) * 8 * T, hipMemcpyDeviceToHost);
    for (int t = 0; t < T; ++t) {
        const double ax = h[2 * t].x, ay = h[2 * t].y, bx = h[2 * t + 1].x, by = h[2 * t + 1].y;
        const double wx = 0.5, wy = -0.8660254037844386;
        const double r[8][2] = {{ax * bx - ay * by, ax * by + ay * bx}, {ax * bx + ay * by, ay * bx - ax * by},
                                {ax - by, ay + bx}, {ax + by, ay - bx}, {ax * wx - ay * wy, ax * wy + ay * wx},
                                {ax + 0.3f * bx, ay + 0.3f * by}, {1.5 * ax, 1.5 * ay}, {ax - bx, ay - by}};
        for (int k = 0; k < 8; ++k) ref[8 * t + k] = make_float2((float)r[k][0], (float)r[k][1]);
    }
    worst = fmax(worst, check("primitives", d, ref));
    return worst < 1e-5 ? 0 : 1;
}
