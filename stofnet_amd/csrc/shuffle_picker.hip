// HBM-bound helpers of the hot path, written as wavefront-shuffle kernels for gfx950:
//
//   sample_shuffle  : SampleShuffle1D.forward (utils/sample_shuffle.py:10-28)
//   pick_maxima     : get_maxima_positions (utils/mask2samples.py:26-34) = nms_1d (:5-11)
//                     + thresholding (:14-23) + nonzero (:32), per row, ordered
//   indices_to_coords: the scatter/zero-pad/divide of mask2coords (:92-112)
#include <hip/hip_runtime.h>
#include "stof_common.h"

namespace {

// ----------------------------------------------------------------------------------
// SampleShuffle1D: out[n, c, w*r + k] = in[n, k*C + c, w]   (C = C_in / r)
// One work-group moves a [r x TW] tile of one (n, c) pair through LDS so that both the
// global reads (rows of `in`, contiguous in w) and the global writes (contiguous in
// w*r + k) are coalesced 4-byte-per-lane streams.
// ----------------------------------------------------------------------------------
constexpr int SHUF_TW = 256;      // w positions per tile
__global__ __launch_bounds__(256) void sample_shuffle_kernel(const float* __restrict__ in,
                                                             float* __restrict__ out, int C, int W, int r,
                                                             int tiles_w) {
    extern __shared__ float tile[];                 // [r][TW + 1]
    const int tid = threadIdx.x;
    const int tw = blockIdx.x % tiles_w;
    const long long nc = blockIdx.x / tiles_w;      // n * C + c
    const long long n = nc / C;
    const int c = (int)(nc - n * C);
    const int w0 = tw * SHUF_TW;
    const int wn = min(SHUF_TW, W - w0);
    const float* src = in + (n * (long long)r * C + c) * W + w0;      // row k at + k*C*W
    for (int k = 0; k < r; ++k)
        if (tid < wn) tile[k * (SHUF_TW + 1) + tid] = src[(long long)k * C * W + tid];
    __syncthreads();
    float* dst = out + (nc * W + w0) * (long long)r;
    const int total = wn * r;
    for (int i = tid; i < total; i += 256) {
        const int w = i / r, k = i - w * r;
        dst[i] = tile[k * (SHUF_TW + 1) + w];
    }
}

// ----------------------------------------------------------------------------------
// pick_maxima: one work-group per row.  The row is walked in chunks of CH samples held
// in LDS with a halo of `half` on both sides; each lane owns a run of 4 consecutive samples.
// A sample is a detection iff
//      s == max(window)  &&  s != 0  &&  s >= cut
// where cut = threshold (threshold mode) or the per-row maximum of the NMS output
// (arg-max mode; computed in a first pass: max over samples that are window maxima, and 0
// if any sample is not one).  Detections are emitted in ascending time order with a
// wave-ballot prefix sum.
// ----------------------------------------------------------------------------------
constexpr int PK_CH = 1024;       // samples per chunk (256 threads x 4)
constexpr int PK_MAXHALF = 64;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(256) void pick_maxima_kernel(const float* __restrict__ scores, int M, int half,
                                                          int has_threshold, float threshold,
                                                          int* __restrict__ counts, int* __restrict__ idx,
                                                          long long idx_cap) {
    __shared__ float buf[PK_CH + 2 * PK_MAXHALF];
    __shared__ float red[4];
    __shared__ int wsum[4];
    __shared__ int sflag[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row = blockIdx.x;
    const float* s = scores + row * (long long)M;
    int* out = idx + row * idx_cap;

    auto load_chunk = [&](int c0) {
        for (int i = tid; i < PK_CH + 2 * half; i += 256) {
            const int t = c0 - half + i;
            buf[i] = (t >= 0 && t < M) ? s[t] : -INFINITY;     // max_pool1d pads with -inf
        }
    };
    // window max for the 4 samples owned by this thread (positions c0 + 4*tid + e)
    auto window_max4 = [&](float wm[4], float v[4]) {
        const int b = 4 * tid;                 // buf index of the window start of sample 0
        float run = -INFINITY;
        // samples b .. b+3 have windows [b, b+2half], ..., [b+3, b+3+2half]; shared core [b+3, b+2half]
        for (int i = b + 3; i <= b + 2 * half; ++i) run = fmaxf(run, buf[i]);
        const float l0 = buf[b], l1 = buf[b + 1], l2 = buf[b + 2];
        const float r1 = buf[b + 2 * half + 1], r2 = buf[b + 2 * half + 2], r3 = buf[b + 2 * half + 3];
        wm[0] = fmaxf(run, fmaxf(l0, fmaxf(l1, l2)));
        wm[1] = fmaxf(run, fmaxf(fmaxf(l1, l2), r1));
        wm[2] = fmaxf(run, fmaxf(l2, fmaxf(r1, r2)));
        wm[3] = fmaxf(run, fmaxf(r1, fmaxf(r2, r3)));
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = buf[b + half + e];
        if (half == 0) {                       // window of one sample: everything is its own maximum
#pragma unroll
            for (int e = 0; e < 4; ++e) wm[e] = v[e];
        }
    };

    float cut = threshold;
    if (!has_threshold) {
        // pass 1: per-row max of the NMS output
        float mx = -INFINITY;
        int any_suppressed = 0;
        for (int c0 = 0; c0 < M; c0 += PK_CH) {
            __syncthreads();
            load_chunk(c0);
            __syncthreads();
            float wm[4], v[4];
            window_max4(wm, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = c0 + 4 * tid + e;
                if (t < M) {
                    if (v[e] == wm[e]) mx = fmaxf(mx, v[e]);
                    else any_suppressed = 1;
                }
            }
        }
        mx = wave_max(mx);
        any_suppressed = __any(any_suppressed);
        if (lane == 0) { red[wave] = mx; sflag[wave] = any_suppressed; }
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (sflag[0] | sflag[1] | sflag[2] | sflag[3]) mx = fmaxf(mx, 0.f);   // suppressed samples are 0
        cut = mx;
    }

    int base = 0;                               // detections emitted so far in this row
    for (int c0 = 0; c0 < M; c0 += PK_CH) {
        __syncthreads();
        load_chunk(c0);
        __syncthreads();
        float wm[4], v[4];
        window_max4(wm, v);
        int hit[4], nh = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = c0 + 4 * tid + e;
            hit[e] = (t < M) && (v[e] == wm[e]) && (v[e] != 0.f) && (v[e] >= cut);
            nh += hit[e];
        }
        // exclusive prefix of nh over the work-group (lane order = time order)
        int incl = nh;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) woff += wsum[w];
            tot += wsum[w];
        }
        int pos = base + woff + incl - nh;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (hit[e]) {
                if (pos < idx_cap) out[pos] = c0 + 4 * tid + e;
                ++pos;
            }
        base += tot;
    }
    if (tid == 0) counts[row] = base;
}

__global__ __launch_bounds__(256) void indices_to_coords_kernel(const int* __restrict__ counts,
                                                                const int* __restrict__ idx, long long idx_cap,
                                                                long long N, long long kmax, float r,
                                                                float* __restrict__ coords) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= N * kmax) return;
    const long long row = i / kmax, j = i - row * kmax;
    float v = 0.f;
    if (j < counts[row] && j < idx_cap) v = (float)idx[row * idx_cap + j];
    coords[i] = v / r;                 // IEEE division, as torch's coords /= upsample_factor
}

}  // namespace

extern "C" int stof_sample_shuffle(const float* in, float* out, int64_t N, int64_t C_in, int64_t W, int32_t r,
                                   void* stream) {
    if (N < 0 || C_in < 0 || W < 0 || r < 1) return STOF_ERR_BAD_ARG;
    if (C_in % r != 0) return STOF_ERR_CHANNELS;
    if (N == 0 || C_in == 0 || W == 0) return STOF_OK;
    if (!in || !out) return STOF_ERR_BAD_ARG;
    if (r > 128) return STOF_ERR_UNSUPPORTED;
    const int64_t C = C_in / r;
    const int64_t tiles_w = (W + SHUF_TW - 1) / SHUF_TW;
    const int64_t blocks = N * C * tiles_w;
    if (blocks > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sample_shuffle_kernel, dim3((unsigned)blocks), dim3(256),
                       (size_t)r * (SHUF_TW + 1) * sizeof(float), static_cast<hipStream_t>(stream),
                       in, out, (int)C, (int)W, (int)r, (int)tiles_w);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_pick_maxima(const float* scores, int64_t N, int64_t M, int32_t window_size,
                                int32_t has_threshold, float threshold, int32_t* counts, int32_t* idx,
                                int64_t idx_cap, void* stream) {
    if (N < 0 || M < 0 || idx_cap < 0 || window_size < 0) return STOF_ERR_BAD_ARG;
    if (N == 0) return STOF_OK;
    if ((!scores && M > 0) || !counts || (!idx && idx_cap > 0)) return STOF_ERR_BAD_ARG;
    const int half = (window_size / 2 * 2 + 1 - 1) / 2;        // utils/mask2samples.py:7-8
    if (half > PK_MAXHALF || M > 0x7fffffffLL || N > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pick_maxima_kernel, dim3((unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       scores, (int)M, half, (int)has_threshold, threshold, counts, idx, (long long)idx_cap);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_indices_to_coords(const int32_t* counts, const int32_t* idx, int64_t idx_cap, int64_t N,
                                      int64_t kmax, float upsample_factor, float* coords, void* stream) {
    if (!counts || !idx || !coords || N < 0 || kmax < 0) return STOF_ERR_BAD_ARG;
    if (N * kmax == 0) return STOF_OK;
    const int64_t blocks = (N * kmax + 255) / 256;
    if (blocks > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(indices_to_coords_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), counts, idx, (long long)idx_cap, (long long)N,
                       (long long)kmax, upsample_factor, coords);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
