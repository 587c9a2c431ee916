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
// T = an unsigned integer type of the element's size (1, 2, 4, 8 or 16 bytes): the operation is a pure permutation of elements
// (utils/sample_shuffle.py:24-27 is view / permute / contiguous on any dtype), so every dtype goes through bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void sample_shuffle_kernel(const T* __restrict__ in, T* __restrict__ out, int C, int W, int r,
                                                             int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
    T* const tile = reinterpret_cast<T*>(tile_raw);   // [r][TW + 1]
    const int tid = threadIdx.x;
    const int tw = blockIdx.x % tiles_w;
    const long long nc = blockIdx.x / tiles_w;      // n * C + c
    const long long n = nc / C;
    const int c = (int)(nc - n * C);
    const int w0 = tw * SHUF_TW;
    const int wn = min(SHUF_TW, W - w0);
    const T* src = in + (n * (long long)r * C + c) * W + w0;          // row k at + k*C*W
    const long long rs = (long long)C * W;
    // reads: four independent row loads in flight per thread before the first LDS store
    int k = 0;
    for (; k + 4 <= r; k += 4) {
        T v0 = T(), v1 = T(), v2 = T(), v3 = T();
        if (tid < wn) { v0 = src[k * rs + tid]; v1 = src[(k + 1) * rs + tid]; v2 = src[(k + 2) * rs + tid]; v3 = src[(k + 3) * rs + tid]; }
        tile[k * (SHUF_TW + 1) + tid] = v0;
        tile[(k + 1) * (SHUF_TW + 1) + tid] = v1;
        tile[(k + 2) * (SHUF_TW + 1) + tid] = v2;
        tile[(k + 3) * (SHUF_TW + 1) + tid] = v3;
    }
    for (; k < r; ++k)
        if (tid < wn) tile[k * (SHUF_TW + 1) + tid] = src[k * rs + tid];
    __syncthreads();
    T* dst = out + (nc * W + w0) * (long long)r;
    const int total = wn * r;
    const float inv_r = 1.0f / (float)r;
    for (int i = tid; i < total; i += 256) {
        // i / r without an integer division: exact for i < 2^22 (float quotient + one correction)
        int w = (int)((float)i * inv_r);
        int kk = i - w * r;
        if (kk < 0) { kk += r; w -= 1; } else if (kk >= r) { kk -= r; w += 1; }
        dst[i] = tile[kk * (SHUF_TW + 1) + w];
    }
}
struct alignas(16) shuf_u128 { unsigned long long lo, hi; };

// ----------------------------------------------------------------------------------
// pick_maxima: one work-group per row.  A sample is a detection iff
//      s == max(window)  &&  s != 0  &&  s >= cut
// with cut = threshold (threshold mode) or the per-row maximum of the NMS output (arg-max mode).
//
// Arg-max mode needs no NMS at all.  Let m = max(row):  the NMS output's maximum is m if m > 0
// (the row maximum always survives its own window), so the detections are exactly the positions
// equal to m; if m == 0 nothing is non-zero; if m < 0 a suppressed sample (value 0) outranks m
// unless every sample is its own window maximum, i.e. the row is constant (or the window is a
// single sample) -- then every position equal to m is a detection (Q5).  So: one streaming pass
// (pick_argmax_kernel below).
//
// Threshold mode streams the row through LDS in 4096-sample chunks (+halo, float4 staging); only
// samples that already pass `s >= th && s != 0` (sparse) pay for the window maximum.  Detections are
// emitted in ascending time order with wave-ballot prefix sums.  (A cache-direct / wave-private
// two-pass variant was measured 2-5x slower at moderate candidate density.)
// ----------------------------------------------------------------------------------
constexpr int PK_MAXHALF = 64;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// One streaming pass: every lane keeps the maximum of the samples it has seen, the position of its first occurrence
// and how often it occurred (a lane's positions ascend), plus the minimum.  After the block reduction the detections
// are the `first` positions of the lanes whose maximum equals the row maximum -- sorted, there are one or a handful --
// unless some lane saw the maximum twice or the row is a negative constant (every position is a detection): those rows
// are listed by an ordered second scan (L2-hot).
__global__ __launch_bounds__(256) void pick_argmax_kernel(const float* __restrict__ scores, int M, int half,
                                                          int* __restrict__ counts, int* __restrict__ idx,
                                                          long long idx_cap) {
    __shared__ float red[2][4];
    __shared__ int redi[2][4];
    __shared__ int hitpos[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row = blockIdx.x;
    const float* s = scores + row * (long long)M;
    int* out = idx + row * idx_cap;
    const bool aligned = ((reinterpret_cast<size_t>(s) & 15) == 0);
    const int nq = (M + 3) >> 2;                                     // 4-sample groups, thread tid takes tid, tid + 256, ...

    auto load_q = [&](int q, float (&v)[4]) {
        const int t0 = 4 * q;
        if (aligned && t0 + 3 < M) {
            const float4 f = *reinterpret_cast<const float4*>(s + t0);
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (t0 + e < M) ? s[t0 + e] : NAN;       // NaN: never greater, equal or smaller
        }
    };
    float lmax = -INFINITY, lmin = INFINITY;
    int first = 0, cnt = 0;
    auto take = [&](int q, const float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e];
            if (x > lmax) { lmax = x; first = 4 * q + e; cnt = 1; }
            else if (x == lmax) ++cnt;
            lmin = fminf(lmin, x);                                    // fminf ignores the NaN padding
        }
    };
    int q = tid;
    for (; q + 768 < nq; q += 1024) {                                 // four 16-byte loads in flight per lane
        float v0[4], v1[4], v2[4], v3[4];
        load_q(q, v0); load_q(q + 256, v1); load_q(q + 512, v2); load_q(q + 768, v3);
        take(q, v0); take(q + 256, v1); take(q + 512, v2); take(q + 768, v3);
    }
    for (; q < nq; q += 256) {
        float v0[4];
        load_q(q, v0);
        take(q, v0);
    }
    float m = wave_max(lmax), lo = wave_min(lmin);
    if (lane == 0) { red[0][wave] = m; red[1][wave] = lo; }
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    lo = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
    // which rows have detections at all (see the header comment)
    const bool emit = (m > 0.f) || (m < 0.f && (lo == m || half == 0));
    const bool hit = emit && cnt > 0 && lmax == m;
    const unsigned long long hm = __ballot(hit);
    int nh = __builtin_popcountll(hm), nc = hit ? cnt : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nc += __shfl_xor(nc, o);
    if (lane == 0) { redi[0][wave] = nh; redi[1][wave] = nc; }
    __syncthreads();
    const int hits = redi[0][0] + redi[0][1] + redi[0][2] + redi[0][3];          // lanes holding the maximum
    const int total = redi[1][0] + redi[1][1] + redi[1][2] + redi[1][3];         // positions holding the maximum
    if (!emit || total == 0) {
        if (tid == 0) counts[row] = 0;
        return;
    }
    if (total == hits) {                                              // every such lane saw it once: `first` is the list
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += redi[0][w];
        const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (hit) hitpos[woff + __builtin_popcountll(hm & lt_mask)] = first;
        __syncthreads();
        if (tid < hits) {                                             // rank sort of <= 256 distinct positions
            const int mine = hitpos[tid];
            int rank = 0;
            for (int k = 0; k < hits; ++k) rank += hitpos[k] < mine;
            if (rank < idx_cap) out[rank] = mine;
        }
        if (tid == 0) counts[row] = hits;
        return;
    }
    // a tie inside one lane's stream or a constant negative row: ordered scan by the first wave
    if (wave != 0) return;
    int nout = 0;
    for (int q0 = 0; q0 < nq; q0 += 64) {
        const int qq = q0 + lane;
        float v[4] = {NAN, NAN, NAN, NAN};
        if (qq < nq) load_q(qq, v);
        int nhq = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) nhq += (v[e] == m);
        int incl = nhq;                                               // exclusive prefix over lanes (lane order = time order)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        int pos = nout + incl - nhq;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (v[e] == m) {
                if (pos < idx_cap) out[pos] = 4 * qq + e;
                ++pos;
            }
        nout += __shfl(incl, 63);
    }
    if (lane == 0) counts[row] = nout;
}

constexpr int PK_CH = 4096;       // samples per chunk in threshold mode (256 threads x 16)
constexpr int PK_PER = PK_CH / 256;
__global__ __launch_bounds__(256) void pick_threshold_kernel(const float* __restrict__ scores, int M, int half,
                                                             float threshold, int* __restrict__ counts,
                                                             int* __restrict__ idx, long long idx_cap) {
    // buf[PK_MAXHALF + i] <-> sample c0 + i, i in [-half, PK_CH + half); max_pool1d pads with -inf
    __shared__ __attribute__((aligned(16))) float buf[PK_CH + 2 * PK_MAXHALF];
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row = blockIdx.x;
    const float* s = scores + row * (long long)M;
    int* out = idx + row * idx_cap;
    const bool aligned = ((reinterpret_cast<size_t>(s) & 15) == 0);
    int base = 0;                               // detections emitted so far in this row
    // A chunk is requested one iteration ahead (16-byte pieces + the two halos in registers) and lands in LDS at the
    // top of its iteration, so the HBM latency of chunk k+1 hides behind the scan of chunk k.
    float4 nxt[PK_CH / 1024];
    float hl = -INFINITY, hr = -INFINITY;
    auto request = [&](int c0) {
#pragma unroll
        for (int q = 0; q < PK_CH / 1024; ++q) {
            const int t = c0 + 4 * (tid + 256 * q);
            if (aligned && t + 3 < M) nxt[q] = *reinterpret_cast<const float4*>(s + t);
            else nxt[q] = make_float4(t < M ? s[t] : -INFINITY, t + 1 < M ? s[t + 1] : -INFINITY,
                                      t + 2 < M ? s[t + 2] : -INFINITY, t + 3 < M ? s[t + 3] : -INFINITY);
        }
        if (tid < half) {                                        // max_pool1d pads with -inf
            const int tl = c0 - half + tid, tr = c0 + PK_CH + tid;
            hl = (tl >= 0) ? s[tl] : -INFINITY;
            hr = (tr < M) ? s[tr] : -INFINITY;
        }
    };
    request(0);
    for (int c0 = 0; c0 < M; c0 += PK_CH) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PK_CH / 1024; ++q) *reinterpret_cast<float4*>(buf + PK_MAXHALF + 4 * (tid + 256 * q)) = nxt[q];
        if (tid < half) {
            buf[PK_MAXHALF - half + tid] = hl;
            buf[PK_MAXHALF + PK_CH + tid] = hr;
        }
        __syncthreads();
        if (c0 + PK_CH < M) request(c0 + PK_CH);
        // thread owns PK_PER consecutive samples c0 + PK_PER*tid + e (thread order = time order).  Candidates (>= threshold,
        // non-zero: `thresholding` zeroes the rest and `nonzero` drops zeros) are found branch-free from four 16-byte LDS
        // reads -- the sample-by-sample loop with its compares and branches was 24 VALU instructions per sample, the
        // kernel's actual bound -- and only they pay for the window maximum.  Samples beyond the row are -inf in `buf`.
        unsigned cand = 0;
        {
            const float4* own = reinterpret_cast<const float4*>(buf + PK_MAXHALF + PK_PER * tid);
#pragma unroll
            for (int q = 0; q < PK_PER / 4; ++q) {
                const float4 v = own[q];
                cand |= (unsigned)(v.x >= threshold && v.x != 0.f) << (4 * q);
                cand |= (unsigned)(v.y >= threshold && v.y != 0.f) << (4 * q + 1);
                cand |= (unsigned)(v.z >= threshold && v.z != 0.f) << (4 * q + 2);
                cand |= (unsigned)(v.w >= threshold && v.w != 0.f) << (4 * q + 3);
            }
        }
        unsigned hitmask = 0;
        while (cand) {
            const int e = __builtin_ctz(cand);
            cand &= cand - 1;
            const int bi = PK_MAXHALF + PK_PER * tid + e;
            const float v = buf[bi];
            float wm = v;
            for (int d = 1; d <= half; ++d) wm = fmaxf(wm, fmaxf(buf[bi - d], buf[bi + d]));
            if (v == wm) hitmask |= 1u << e;
        }
        const int nh = __builtin_popcount(hitmask);
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
        while (hitmask) {
            const int e = __builtin_ctz(hitmask);
            hitmask &= hitmask - 1;
            if (pos < idx_cap) out[pos] = c0 + PK_PER * tid + e;
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

// mask2coords' echo_max reduction (utils/mask2samples.py:105-107 -> reduce_echoes :117-132, get_amplitudes :135-136)
// for rows with more than echo_max detections batch-wide: every row's kmax entries (its detections, then the
// reference's zero padding, whose "amplitude" is scores[row, 0]) are ranked by amplitude, the echo_max largest kept
// (ties: the earlier entry) and written in ascending coordinate order, i.e. kept padding first, then the kept
// detections in time order; coordinates are divided by the upsample factor (:112).  One wavefront per row.
__global__ __launch_bounds__(256) void reduce_echoes_kernel(const float* __restrict__ scores, long long M,
                                                            const int* __restrict__ counts, const int* __restrict__ idx,
                                                            long long idx_cap, long long N, int kmax, int k, float r,
                                                            float* __restrict__ coords) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* s = scores + row * M;
    const int cnt = min(counts[row], kmax);
    const int nt = (kmax + 63) / 64;                       // entries lane + 64 t, t < nt <= 64 (kmax <= 4096)
    auto coord_of = [&](int e) { return e < cnt ? idx[row * idx_cap + e] : 0; };
    unsigned long long taken = 0;
    for (int round = 0; round < k; ++round) {
        float best = -INFINITY;
        int best_e = 0x7fffffff;
        for (int t = 0; t < nt; ++t) {
            const int e = lane + 64 * t;
            if (e < kmax && !((taken >> t) & 1ull)) {
                const float a = s[coord_of(e)];
                if (a > best || best_e == 0x7fffffff) { best = a; best_e = e; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oe = __shfl_xor(best_e, o);
            if (oe != 0x7fffffff && (best_e == 0x7fffffff || ob > best || (ob == best && oe < best_e))) { best = ob; best_e = oe; }
        }
        if (best_e != 0x7fffffff && (best_e & 63) == lane) taken |= 1ull << (best_e >> 6);
    }
    // kept padding entries (coordinate 0) sort in front of the kept detections
    int npad = 0;
    for (int t = 0; t < nt; ++t) {
        const int e = lane + 64 * t;
        npad += __builtin_popcountll(__ballot(((taken >> t) & 1ull) && e >= cnt));
    }
    float* out = coords + row * (long long)k;
    for (int p = lane; p < npad; p += 64) out[p] = 0.f / r;
    int pos = npad;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int t = 0; t < nt; ++t) {
        const int e = lane + 64 * t;
        const bool sel = ((taken >> t) & 1ull) && e < cnt;
        const unsigned long long sm = __ballot(sel);
        if (sel) out[pos + __builtin_popcountll(sm & lt_mask)] = (float)coord_of(e) / r;
        pos += __builtin_popcountll(sm);
    }
}

}  // namespace

extern "C" int stof_reduce_echoes(const float* scores, int64_t N, int64_t M, const int32_t* counts, const int32_t* idx,
                                  int64_t idx_cap, int64_t kmax, int64_t echo_max, float upsample_factor, float* coords,
                                  void* stream) {
    if (!scores || !counts || !idx || !coords || N < 0 || M < 1 || kmax < 1 || echo_max < 1 || echo_max >= kmax || kmax > idx_cap)
        return STOF_ERR_BAD_ARG;
    if (N == 0) return STOF_OK;
    if (kmax > 4096 || N > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(reduce_echoes_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       scores, (long long)M, counts, idx, (long long)idx_cap, (long long)N, (int)kmax, (int)echo_max,
                       upsample_factor, coords);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

namespace {
template <typename T>
int launch_shuffle(const void* in, void* out, int64_t N, int64_t C_in, int64_t W, int32_t r, void* stream) {
    const int64_t C = C_in / r;
    const int64_t tiles_w = (W + SHUF_TW - 1) / SHUF_TW;
    const int64_t blocks = N * C * tiles_w;
    if (blocks > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sample_shuffle_kernel<T>, dim3((unsigned)blocks), dim3(256), (size_t)r * (SHUF_TW + 1) * sizeof(T),
                       static_cast<hipStream_t>(stream), static_cast<const T*>(in), static_cast<T*>(out), (int)C, (int)W, (int)r,
                       (int)tiles_w);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
}  // namespace

extern "C" int stof_sample_shuffle_bytes(const void* in, void* out, int64_t N, int64_t C_in, int64_t W, int32_t r,
                                         int32_t elem_bytes, void* stream) {
    if (N < 0 || C_in < 0 || W < 0 || r < 1) return STOF_ERR_BAD_ARG;
    if (C_in % r != 0) return STOF_ERR_CHANNELS;
    if (N == 0 || C_in == 0 || W == 0) return STOF_OK;
    if (!in || !out) return STOF_ERR_BAD_ARG;
    if (r > 128 || (elem_bytes == 16 && r > 32)) return STOF_ERR_UNSUPPORTED;      // the [r][257] LDS tile
    switch (elem_bytes) {
    case 1: return launch_shuffle<unsigned char>(in, out, N, C_in, W, r, stream);
    case 2: return launch_shuffle<unsigned short>(in, out, N, C_in, W, r, stream);
    case 4: return launch_shuffle<unsigned int>(in, out, N, C_in, W, r, stream);
    case 8: return launch_shuffle<unsigned long long>(in, out, N, C_in, W, r, stream);
    case 16: return launch_shuffle<shuf_u128>(in, out, N, C_in, W, r, stream);
    default: return STOF_ERR_UNSUPPORTED;
    }
}

extern "C" int stof_sample_shuffle(const float* in, float* out, int64_t N, int64_t C_in, int64_t W, int32_t r,
                                   void* stream) {
    return stof_sample_shuffle_bytes(in, out, N, C_in, W, r, 4, stream);
}

extern "C" int stof_pick_maxima(const float* scores, int64_t N, int64_t M, int32_t window_size,
                                int32_t has_threshold, float threshold, int32_t* counts, int32_t* idx,
                                int64_t idx_cap, void* stream) {
    if (N < 0 || M < 0 || idx_cap < 0 || window_size < 0) return STOF_ERR_BAD_ARG;
    if (N == 0) return STOF_OK;
    if ((!scores && M > 0) || !counts || (!idx && idx_cap > 0)) return STOF_ERR_BAD_ARG;
    const int half = (window_size / 2 * 2 + 1 - 1) / 2;        // utils/mask2samples.py:7-8
    if (half > PK_MAXHALF || M > 0x7fffffffLL || N > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    if (has_threshold)
        hipLaunchKernelGGL(pick_threshold_kernel, dim3((unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream),
                           scores, (int)M, half, threshold, counts, idx, (long long)idx_cap);
    else
        hipLaunchKernelGGL(pick_argmax_kernel, dim3((unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream),
                           scores, (int)M, half, counts, idx, (long long)idx_cap);
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
