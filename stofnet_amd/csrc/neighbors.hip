// The steps on either side of the hot path:
//
//   iq2rf_kernel   : datasets/chirp_dataset.py:80-91 `iq2rf` -- linear resampling of the complex IQ
//                    trace to int(len * rf) points on endpoint-inclusive grids, up-mixing with
//                    exp(2 pi i fc t), real part -- followed by NormalizeVol (utils/transforms.py:13),
//                    x / max|x| per waveform.  One work-group per waveform, HBM-bound.
//   toa_rmse_kernel: utils/metrics.py:9-41 -- per row nearest squared distance GT x EST, tolerance
//                    gate, RMSE / precision / recall / Jaccard / TP / FP / FN.  One wavefront per row.
#include <hip/hip_runtime.h>
#include "stof_common.h"

namespace {

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// iq[n][len][2] (re, im) fp32 -> rf[n][M] fp32, M = int(len * rescale).  The reference computes in
// float64 and casts to float32 (main.py:305); here the interpolation weight comes from exact
// integer arithmetic and the carrier phase fc*t is reduced mod 1 in float64 before the fp32 sincos,
// so the result agrees to fp32 rounding.
__global__ __launch_bounds__(256) void iq2rf_kernel(const float* __restrict__ iq, float* __restrict__ rf,
                                                    int len, int M, double fc, double fs, int normalize) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row = blockIdx.x;
    const float2* src = reinterpret_cast<const float2*>(iq) + row * (size_t)len;
    float* dst = rf + row * (size_t)M;
    const double T = (double)len / fs;                     // np.linspace(0, len/fs, ..., endpoint=True)
    const double dt = M > 1 ? T / (double)(M - 1) : 0.0;
    float amax = 0.f;
    for (int j = tid; j < M; j += 256) {
        // u = j * (len-1) / (M-1): integer part and exact fractional weight
        const long long num = (long long)j * (len - 1);
        const int den = M > 1 ? M - 1 : 1;
        int i0 = (int)(num / den);
        float w = (float)((double)(num - (long long)i0 * den) / (double)den);
        if (i0 >= len - 1) { i0 = max(len - 2, 0); w = len > 1 ? 1.f : 0.f; }
        const float2 a = src[i0];
        const float2 b = src[min(i0 + 1, len - 1)];
        const float yr = fmaf(w, b.x - a.x, a.x), yi = fmaf(w, b.y - a.y, a.y);
        const double cyc = fc * ((double)j * dt);          // carrier cycles at t_j
        const float fr = (float)(cyc - floor(cyc));
        float sn, cs;
        sincospif(2.0f * fr, &sn, &cs);
        const float v = yr * cs - yi * sn;                 // Re{ y * exp(2 pi i fc t) }
        dst[j] = v;
        amax = fmaxf(amax, fabsf(v));
    }
    if (!normalize) return;
    amax = wave_max_f(amax);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int j = tid; j < M; j += 256) dst[j] = dst[j] / amax;      // NormalizeVol: waveform / abs(waveform).max()
}

// gt[N][G], es[N][E] fp32 (0 / NaN / inf entries are padding, utils/metrics.py:6) -> out[N][7]
__global__ __launch_bounds__(256) void toa_rmse_kernel(const float* __restrict__ gt, const float* __restrict__ es,
                                                       int N, int G, int E, float tol, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* g = gt + row * (long long)G;
    const float* e = es + row * (long long)E;
    auto ok = [](float v) { return v != 0.f && !isnan(v) && !isinf(v); };
    int nes = 0;
    for (int k = lane; k < E; k += 64) nes += ok(e[k]) ? 1 : 0;
    int ngt = 0;
    float sum = 0.f;
    int tp = 0, fn = 0;
    for (int i = lane; i < G; i += 64) {
        const float gv = g[i];
        if (!ok(gv)) continue;
        ++ngt;
        float best = INFINITY;
        for (int k = 0; k < E; ++k) {
            const float ev = e[k];
            if (ok(ev)) { const float d = gv - ev; best = fminf(best, d * d); }
        }
        if (best <= tol) { sum += best; ++tp; } else { ++fn; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nes += __shfl_xor(nes, o); ngt += __shfl_xor(ngt, o);
        tp += __shfl_xor(tp, o); fn += __shfl_xor(fn, o);
        sum += __shfl_xor(sum, o);
    }
    if (lane != 0) return;
    float* o7 = out + row * 7;
    float mes = 0.f, tps = 0.f, fps = 0.f, fns = 0.f;
    if (ngt > 0 && nes > 0) {            // utils/metrics.py:28-29: rows with no valid GT or estimate stay zero
        mes = sqrtf(sum / (float)tp);    // mean of an empty selection is NaN, as torch.mean
        tps = (float)tp; fns = (float)fn; fps = (float)nes - tps;
    }
    o7[0] = mes;
    o7[1] = tps / (fps + tps) * 100.f;
    o7[2] = tps / (fns + tps) * 100.f;
    o7[3] = tps / (fns + tps + fps) * 100.f;
    o7[4] = tps; o7[5] = fps; o7[6] = fns;
}

}  // namespace

extern "C" int stof_iq2rf(const float* iq, float* rf, int64_t N, int64_t len, double rescale_factor, double fc,
                          double fs, int32_t normalize, void* stream) {
    if (N < 0 || len < 0 || !(rescale_factor > 0) || !(fs > 0)) return STOF_ERR_BAD_ARG;
    const int64_t M = (int64_t)((double)len * rescale_factor);      // int(data_len * rescale_factor)
    if (N == 0 || len == 0 || M == 0) return STOF_OK;
    if (!iq || !rf) return STOF_ERR_BAD_ARG;
    if (N > 0x7fffffffLL || M > 0x7fffffffLL || len > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(iq2rf_kernel, dim3((unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream), iq, rf,
                       (int)len, (int)M, fc, fs, (int)normalize);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_toa_rmse(const float* gt, const float* es, int64_t N, int64_t G, int64_t E, float tol,
                             float* out, void* stream) {
    if (N < 0 || G < 0 || E < 0) return STOF_ERR_BAD_ARG;
    if (N == 0) return STOF_OK;
    if (!out || (!gt && G > 0) || (!es && E > 0)) return STOF_ERR_BAD_ARG;
    if (N > 0x7fffffffLL || G > 0x7fffffffLL || E > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(toa_rmse_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       gt, es, (int)N, (int)G, (int)E, tol, out);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
