// GradPeak (models/gradpeak.py:8-133) on gfx950.
//
//   gradpeak_rows_kernel<MOMENTS>   grad_peak_detect on envelope rows in HBM, one wavefront per row (gradpeak_core.h's block
//                                   streamer): MOMENTS = true is the pre-pass of the default threshold (Q7: the std of the
//                                   WHOLE batch of smoothed gradients, :18), MOMENTS = false detects and pairs
//   gradpeak_split_kernel<MOMENTS>  the same for few long rows: 1, 2 or 4 waves share a row's iterations
//   gradpeak_threshold_kernel       thres_pos = std**16 * 1.2e13 from the (all-reduced) moments, on the device
//   toa_fused_ct_kernel<N, ...>,    toa_detect (:99-116) from the waveforms with the envelope kept in LDS: a pair of rows
//   toa_fused_kernel                goes through FFT -> Hilbert filter -> inverse FFT (compile-time plan for 1536 / 2000 /
//                                   2048 samples, run-time plan otherwise), becomes two envelopes in place and is streamed
//                                   through gradient -> Gaussian blur -> threshold crossings -> hysteresis pairing ->
//                                   echo_max reduction (explicit threshold, one launch), or through the moments pre-pass
//                                   (MOMENTS: the envelope is also stored for the detection launch)
//   gradpeak_flags_kernel           detection from a smoothed gradient kept by stof_gradpeak_moments_store (r2 route, ABI only)
// One host read per call at most (flags = {Q9, Kmax}); no host sync inside.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "stof_common.h"
#include "stof_hip_util.h"
#include "fft_small.h"
#include "pair_io.h"
#include "ct_twiddles.h"
#include "gradpeak_core.h"

namespace stof {
size_t hilbert_fast_lds_bytes(int64_t n, stof_fft::Plan* plan_out);      // hilbert.hip
}

namespace {

using stof_gp::Config;
using stof_gp::RowState;
constexpr int LDS_BYTES = 160 * 1024;
// waves (= rows in flight) per work-group of gradpeak_rows_kernel: 4 when detecting; 16 for the moments, whose work-groups
// end in two double-precision atomic adds on ONE pair of addresses (~90 atomics per microsecond: 1024 small groups spent
// 20 us of a 40 us kernel queueing there; 256 groups of 16 waves do not)
constexpr int ROWS_WAVES = 4, MOMENT_WAVES = 16;

// Dynamic LDS: a gradient buffer of ring_stride floats per wave.  (Copying the wave's whole row into LDS first, every
// 16-byte load in flight at once, was measured and dropped: [4096, 2000] 29.9 us against 26.9 us for streaming straight
// from HBM with the next iteration's samples requested one iteration ahead -- the copy is a phase nothing overlaps.)
template <bool MOMENTS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gradpeak_rows_kernel(const float* __restrict__ env, long long N, Config cf,
                                                                        const float* __restrict__ taps,
                                                                        const float* __restrict__ th_dev,
                                                                        float* __restrict__ echoes, float* __restrict__ reduced,
                                                                        int* __restrict__ counts, int* __restrict__ flags,
                                                                        double* __restrict__ stats, float* __restrict__ blurred,
                                                                        int ring_stride) {
    __shared__ __attribute__((aligned(16))) float tp[stof_gp::TAPS_LDS];
    __shared__ double red[2][WAVES];
    extern __shared__ __attribute__((aligned(16))) float rows_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const ring = rows_lds + (size_t)wave * ring_stride;
    stof_gp::stage_taps(tp, taps, cf.radius, tid, blockDim.x);
    if (th_dev != nullptr) {                                   // default threshold computed on the device (Q7)
        cf.th_pos = *th_dev;
        cf.th_neg = -cf.th_pos / 4.0f;                          // models/gradpeak.py:19
    }
    __syncthreads();
    double mom[2] = {0.0, 0.0};
    int kmax = 0;                                              // largest echo count among this wave's rows
    for (long long row = (long long)blockIdx.x * WAVES + wave; row < N; row += (long long)gridDim.x * WAVES) {
        const float* e = env + row * (long long)cf.L;
        float* const out[1] = {MOMENTS ? nullptr : echoes + row * cf.cap * 3};
        RowState st[1];
        if (MOMENTS && blurred != nullptr) {                   // keep the smoothed gradient for stof_grad_peak_detect_blurred
            const int nw = stof_gp::word_count(cf.L, cf.radius);
            float* const b = blurred + row * (long long)nw * 64;
            stof_gp::stream_words<1>(cf, tp, ring, lane, [&](int u, float (&v)[1]) { v[0] = e[u]; }, 0, nw - 1, true,
                                     [&](int c, int, unsigned long long, unsigned long long, unsigned long long, float sm) {
                                         mom[0] += (double)sm;
                                         mom[1] += (double)sm * (double)sm;
                                         b[64 * c + lane] = sm;
                                     });
            continue;
        }
        if constexpr (MOMENTS) {
            stof_gp::moments_row_blocks(cf, tp, ring, lane, stof_gp::EnvRow{e}, mom, taps);
        } else {
            stof_gp::detect_row_blocks(cf, tp, ring, lane, stof_gp::EnvRow{e}, out[0], st[0], taps);
            stof_gp::finish_row(st[0], cf, row, out[0], reduced, counts, flags, lane, true);
            kmax = st[0].nout > kmax ? st[0].nout : kmax;
        }
    }
    if (!MOMENTS) {                                            // Kmax of the batch: one atomic per work-group, not per row
        int* const wmax = reinterpret_cast<int*>(&red[0][0]);
        if (lane == 0) wmax[wave] = kmax;
        __syncthreads();
        if (tid == 0) {
            int m = 0;
            for (int w = 0; w < WAVES; ++w) m = wmax[w] > m ? wmax[w] : m;
            if (m > 0 && m > __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&flags[1], m);
        }
    }
    if (MOMENTS) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mom[0] += __shfl_xor(mom[0], o);
            mom[1] += __shfl_xor(mom[1], o);
        }
        if (lane == 0) { red[0][wave] = mom[0]; red[1][wave] = mom[1]; }
        __syncthreads();
        if (tid == 0) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < WAVES; ++w) { a += red[0][w]; b += red[1][w]; }
            atomicAdd(&stats[0], a);
            atomicAdd(&stats[1], b);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// grad_peak_detect on envelope rows in HBM, work-group of 4 waves: each row is streamed by W = 1, 2 or 4 waves that
// split its iterations (a wave that starts inside the row runs one extra iteration to fill its gradient history), so
// few long rows still fill the chip.  The flag words of the row go to LDS; after a barrier the first wave of the row
// pairs them (pair_stored_words).  MOMENTS: the sums of the default threshold (Q7).
// ----------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int split_flag_offset(int radius) {   // floats before the flag words (8-byte aligned)
    const int f = stof_gp::TAPS_LDS + 4 * stof_gp::block_buf_floats(radius);
    return f + (f & 1);
}
inline size_t split_lds_bytes(int L, int radius, int W) {
    // flag words (P, M, V) of every iteration's four words, for the rows of one work-group; the MOMENTS variant reuses
    // the region for 8 doubles
    size_t flag_bytes = (size_t)(4 / W) * 3 * stof_gp::block_iterations(L, radius) * stof_gp::WPI * 8;
    if (flag_bytes < 64) flag_bytes = 64;
    return (size_t)split_flag_offset(radius) * 4 + flag_bytes;
}

template <bool MOMENTS>
__global__ __launch_bounds__(256) void gradpeak_split_kernel(const float* __restrict__ env, long long N, Config cf,
                                                             const float* __restrict__ taps, const float* __restrict__ th_dev,
                                                             float* __restrict__ echoes, float* __restrict__ reduced,
                                                             int* __restrict__ counts, int* __restrict__ flags,
                                                             double* __restrict__ stats, int W) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = cf.L, rad = cf.radius, nwords = stof_gp::word_count(L, rad), niter = stof_gp::block_iterations(L, rad), RPW = 4 / W;
    float* const tp = lds_f;
    float* const ring = tp + stof_gp::TAPS_LDS + wave * stof_gp::block_buf_floats(rad);
    stof_gp::stage_taps(tp, taps, rad, tid, 256);
    if (th_dev != nullptr) {                                   // default threshold computed on the device (Q7)
        cf.th_pos = *th_dev;
        cf.th_neg = -cf.th_pos / 4.0f;                          // models/gradpeak.py:19
    }
    const int slot = wave / W, part = wave % W;                  // row slot of this wave, its share of the row's iterations
    unsigned long long* const F = reinterpret_cast<unsigned long long*>(tp + split_flag_offset(rad)) + (size_t)slot * 3 * niter * stof_gp::WPI;
    // iterations [it_b, it_e) of this wave (contiguous, near-equal shares)
    const int per = (niter + W - 1) / W, it_b = part * per, it_e = it_b + per < niter ? it_b + per : niter;
    double mom[2] = {0.0, 0.0};
    __syncthreads();
    for (long long row0 = (long long)blockIdx.x * RPW; row0 < N; row0 += (long long)gridDim.x * RPW) {
        const long long row = row0 + slot;
        const float* const e = env + row * (long long)L;
        if (row < N && it_b < it_e) {
            if constexpr (MOMENTS) {
                stof_gp::stream_blocks(cf, tp, ring, lane, stof_gp::EnvRow{e},
                                       [&](int, const unsigned long long (&)[stof_gp::WPI], const unsigned long long (&)[stof_gp::WPI],
                                           const float (&sm)[stof_gp::WPI]) {
#pragma unroll
                                           for (int k = 0; k < stof_gp::WPI; ++k) {
                                               mom[0] += (double)sm[k];
                                               mom[1] += (double)sm[k] * (double)sm[k];
                                           }
                                       },
                                       [](int) {}, taps, it_b, it_e);
            } else {
                stof_gp::FlagBatch fb;
                const int w_b = stof_gp::WPI * it_b, w_e = stof_gp::WPI * it_e;
                stof_gp::stream_blocks(cf, tp, ring, lane, stof_gp::EnvRow{e},
                                       [&](int c, const unsigned long long (&P)[stof_gp::WPI], const unsigned long long (&M)[stof_gp::WPI],
                                           const float (&)[stof_gp::WPI]) {
#pragma unroll
                                           for (int k = 0; k < stof_gp::WPI; ++k) fb.put(lane, (c + k) & 63, P[k], M[k]);
                                       },
                                       [&](int next) {             // 64 words collected (or the share is done): lane l stores word c0 + l
                                           if ((next & 63) == 0 || next >= w_e) {
                                               const int c = ((next - 1) & ~63) + lane;
                                               if (c >= w_b && c < next) {
                                                   F[3 * c] = ((unsigned long long)fb.phi << 32) | fb.plo;
                                                   F[3 * c + 1] = ((unsigned long long)fb.mhi << 32) | fb.mlo;
                                                   F[3 * c + 2] = stof_gp::valid_word(c, L, rad);
                                               }
                                           }
                                       },
                                       taps, it_b, it_e);
            }
        }
        if (!MOMENTS) {
            __syncthreads();                                      // the row's flag words are complete
            if (row < N && part == 0) {
                float* const out = echoes + row * cf.cap * 3;
                RowState st;
                stof_gp::pair_stored_words(st, F, nwords, lane, cf, out, [&](int i) { return e[i]; });
                stof_gp::finish_row(st, cf, row, out, reduced, counts, flags, lane);
            }
            __syncthreads();                                      // before the next rows overwrite the flag words
        }
    }
    if (MOMENTS) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mom[0] += __shfl_xor(mom[0], o);
            mom[1] += __shfl_xor(mom[1], o);
        }
        double* const red = reinterpret_cast<double*>(tp + split_flag_offset(rad));      // the flag words are unused in this mode
        if (lane == 0) { red[wave] = mom[0]; red[4 + wave] = mom[1]; }
        __syncthreads();
        if (tid == 0) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < 4; ++w) { a += red[w]; b += red[4 + w]; }
            atomicAdd(&stats[0], a);
            atomicAdd(&stats[1], b);
        }
    }
}

// waves per row: one while the rows alone fill the chip (16 waves per CU), more for fewer rows
int split_waves_per_row(int64_t N, int ncu) {
    if (N >= (int64_t)ncu * 12) return 1;
    if (N >= (int64_t)ncu * 6) return 2;
    return 4;
}

// launches the split kernel if its LDS fits; false: the caller uses gradpeak_rows_kernel
template <bool MOMENTS>
bool launch_split(const float* env, int64_t N, const Config& cf, const float* taps, const float* th_dev, float* echoes,
                  float* reduced, int* counts, int* flags, double* stats, hipStream_t stream, int* status) {
    // STOF_GP_SPLIT (read per call: tests flip it): 0 = never, 1 / unset = by shape, 2 / 3 / 4 = always with 1 / 2 / 4 waves per row
    const char* const e = getenv("STOF_GP_SPLIT");
    const int mode = e ? atoi(e) : 1;
    if (mode == 0) return false;
    const int ncu = stof::device_cu_count();
    // Measured on the MI355X: with enough rows to give every CU 12+ waves (one per row), gradpeak_rows_kernel is the
    // faster one ([4096,2000]: 72 vs 88 us -- 16 waves per CU hide its per-word HBM latency and it has no barriers);
    // few long rows are what this kernel is for ([512,30720]: 96 vs 278 us).
    if (mode == 1 && N >= (int64_t)ncu * 12) return false;
    int W = mode > 1 ? (mode == 2 ? 1 : (mode == 3 ? 2 : 4)) : split_waves_per_row(N, ncu);
    size_t lds = split_lds_bytes(cf.L, cf.radius, W);
    while (lds > (size_t)LDS_BYTES && W < 4) { W *= 2; lds = split_lds_bytes(cf.L, cf.radius, W); }
    if (lds > (size_t)LDS_BYTES) return false;
    static stof::LdsLimitOnce once;
    if (int st = once.ensure(reinterpret_cast<const void*>(&gradpeak_split_kernel<MOMENTS>), LDS_BYTES)) { *status = st; return true; }
    const int RPW = 4 / W;
    int64_t grid = (N + RPW - 1) / RPW;
    int64_t per_cu = (int64_t)((size_t)LDS_BYTES / lds);
    if (per_cu > 8) per_cu = 8;
    if (grid > (int64_t)ncu * per_cu) grid = (int64_t)ncu * per_cu;
    hipLaunchKernelGGL(gradpeak_split_kernel<MOMENTS>, dim3((unsigned)grid), dim3(256), lds, stream, env, (long long)N, cf, taps,
                       th_dev, echoes, reduced, counts, flags, stats, W);
    *status = hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
    return true;
}

// Detection from the smoothed gradient kept by the moments pass (row-major, 64 values per iteration): the loads do not
// depend on anything computed here, so several iterations are requested ahead; flags, edges and pairing word by word.
__global__ __launch_bounds__(64 * ROWS_WAVES) void gradpeak_flags_kernel(const float* __restrict__ env, const float* __restrict__ blurred,
                                                                         long long N, Config cf, const float* __restrict__ th_dev,
                                                                         float* __restrict__ echoes, float* __restrict__ reduced,
                                                                         int* __restrict__ counts, int* __restrict__ flags) {
    __shared__ int wmax[ROWS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (th_dev != nullptr) {
        cf.th_pos = *th_dev;
        cf.th_neg = -cf.th_pos / 4.0f;                          // models/gradpeak.py:19
    }
    const int L = cf.L, rad = cf.radius, nw = stof_gp::word_count(L, rad);
    constexpr int AHEAD = 8;
    int kmax = 0;
    for (long long row = (long long)blockIdx.x * ROWS_WAVES + wave; row < N; row += (long long)gridDim.x * ROWS_WAVES) {
        const float* const b = blurred + row * (long long)nw * 64 + lane;
        const float* const e = env + row * (long long)L;
        float* const out = echoes + row * cf.cap * 3;
        RowState st;
        for (int c0 = 0; c0 < nw; c0 += AHEAD) {
            float sm[AHEAD];
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) sm[k] = (c0 + k < nw) ? b[64 * (c0 + k)] : 0.f;
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                const int c = c0 + k;
                if (c >= nw) break;
                const int i = 64 * c + lane - rad;
                const bool in_row = (i >= 0) && (i < L);
                const unsigned long long V = __ballot(in_row && i < L - 1);
                const unsigned long long P = __ballot(in_row && sm[k] > cf.th_pos);
                const unsigned long long M = __ballot(in_row && sm[k] < cf.th_neg);
                if (c > 0) {
                    const unsigned long long EP = ~st.P & ((st.P >> 1) | (P << 63)) & st.V;
                    const unsigned long long EM = ~st.M & ((st.M >> 1) | (M << 63)) & st.V;
                    stof_gp::pair_word(st, 64 * (c - 1) - rad, EP, EM, lane, cf, out, [&](int idx) { return e[idx]; });
                }
                st.P = P; st.M = M; st.V = V;
            }
        }
        stof_gp::finish_row(st, cf, row, out, reduced, counts, flags, lane, true);
        kmax = st.nout > kmax ? st.nout : kmax;
    }
    if (lane == 0) wmax[wave] = kmax;
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int w = 0; w < ROWS_WAVES; ++w) m = wmax[w] > m ? wmax[w] : m;
        if (m > 0 && m > __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&flags[1], m);
    }
}

// thres_pos = (grad_data.std() ** 16) * 1.2e13 (models/gradpeak.py:18): unbiased std of all N * L smoothed gradients
// from stats = (sum, sum of squares, count), rounded to fp32 where torch rounds (std is an fp32 tensor, the power and
// the product are fp32 operations); the 16th power is taken in double and rounded once (= a correctly rounded powf).
__global__ void gradpeak_threshold_kernel(const double* __restrict__ stats, float* __restrict__ th_out) {
    const double s1 = stats[0], s2 = stats[1], cnt = stats[2];
    double var = (s2 - s1 * s1 / cnt) / (cnt > 1.0 ? cnt - 1.0 : 1.0);
    if (!(var > 0.0)) var = 0.0;
    const float sd = (float)sqrt(var);
    double p = (double)sd;
    p *= p; p *= p; p *= p; p *= p;                            // ** 16
    const float p16 = (float)p;
    th_out[0] = p16 * 1.2e13f;
}

// ----------------------------------------------------------------------------------------------------------------
// toa_detect fused: one wavefront per pair of waveforms, no work-group barrier after the table set-up
// ----------------------------------------------------------------------------------------------------------------
struct FusedParams {
    const float* x;        // [N][L] waveforms
    const float* taps;     // [2 rad + 1]
    float* echoes;         // [N][cap][3]
    float* reduced;        // [N][echo_max][3] or nullptr
    int* counts;           // [N]
    int* flags;            // {Q9, Kmax}
    float* env_out;        // optional [N][L]: also store the envelope (the moments pass keeps it for the detection launch)
    double* partials;      // MOMENTS: [STOF_MOMENT_SLOTS][16] doubles; a work-group adds {sum, sum of squares} of its smoothed
                           // gradients to slot blockIdx & (SLOTS - 1), entries 0 and 1 (one 128-byte line per slot)
    long long N;
    Config cf;
    stof_fft::Plan plan;
};

#ifdef STOF_GP_STAMPS                                              // diagnostic build: cycles per phase of waves 0 and 1 of each group
constexpr int GP_STAMP_SLOTS = 8192;
__device__ unsigned long long g_gp_stamps[GP_STAMP_SLOTS][8];
#define GP_STAMP(i) do { if (lane == 0 && wave < 2) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         g_gp_stamps[(2 * blockIdx.x + wave) % GP_STAMP_SLOTS][i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define GP_STAMP(i) do { } while (0)
#endif

// One work-group (T = 64, 128 or 256 threads) per pair of waveforms; after the transform wave r of the group streams
// row r of the pair (a single-wave group takes both rows one after the other).  MOMENTS: the pre-pass of the default
// threshold (Q7) -- the streaming stage sums the smoothed gradient instead of detecting.
template <bool MOMENTS>
__global__ __launch_bounds__(256, 4) void toa_fused_kernel(const FusedParams p) {
    using namespace stof_fft;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int n = p.cf.L, tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    // LDS: pair image [n] | twiddle tables | tap image | two gradient buffers
    cf* const Z = reinterpret_cast<cf*>(lds);
    cf* const ta = Z + n;
    cf* const tb = ta + TW_A;
    const int nb = (n + TW_A - 1) / TW_A;
    float* const tp = reinterpret_cast<float*>(tb + nb + (nb & 1));              // 16-byte aligned (tables are 8-byte entries)
    float* const rings = tp + stof_gp::TAPS_LDS;
#ifdef STOF_GP_STAMPS
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
    stof_gp::stage_taps(tp, p.taps, p.cf.radius, tid, T);
    for (int t = tid; t < TW_A + nb; t += T) {
        const double k = t < TW_A ? (double)t : (double)(t - TW_A) * (double)TW_A;
        double sn, cs;
        sincospi(-2.0 * k / (double)n, &sn, &cs);
        ta[t] = mk((float)cs, (float)sn);
    }
    const Twiddles tw{ta, tb};
    const long long npairs = (p.N + 1) / 2;
    double mom[2] = {0.0, 0.0};
    for (long long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const long long row = 2 * pr;
        const float* xr = p.x + row * (size_t)n;
        const bool second = row + 1 < p.N;
        const float* const x2 = second ? xr + n : nullptr;
        __syncthreads();                                          // tables built / previous pair fully consumed
        GP_STAMP(5);
        stof_io::load_pair<4>(Z, xr, x2, n, tid, T);
        __syncthreads();
        GP_STAMP(0);
        analytic_in_place(Z, p.plan, tw, tid, T, [] { __syncthreads(); });
        GP_STAMP(1);
        float* const eo = p.env_out ? p.env_out + row * (size_t)n : nullptr;
        stof_io::unmix_pair<4>(                                   // analytic signals -> the two envelopes, in place
            Z, xr, x2, n, tid, T,
            [&](int q, const float (&xa)[4], const float (&v1)[4], const float (&xb)[4], const float (&v2)[4]) {
                float ea[4], eb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { ea[e] = stof_io::envelope(xa[e], v1[e]); eb[e] = stof_io::envelope(xb[e], v2[e]); }
                stof_io::f4a* d = reinterpret_cast<stof_io::f4a*>(Z + 4 * q);
                d[0] = make_float4(ea[0], eb[0], ea[1], eb[1]);
                d[1] = make_float4(ea[2], eb[2], ea[3], eb[3]);
                if (eo) {
                    *reinterpret_cast<float4*>(eo + 4 * q) = make_float4(ea[0], ea[1], ea[2], ea[3]);
                    if (second) *reinterpret_cast<float4*>(eo + n + 4 * q) = make_float4(eb[0], eb[1], eb[2], eb[3]);
                }
            },
            [&](int i, float xa, float v1, float xb, float v2) {
                const cf e = mk(stof_io::envelope(xa, v1), stof_io::envelope(xb, v2));
                Z[i] = e;
                if (eo) { eo[i] = e.x; if (second) eo[n + i] = e.y; }
            });
        __syncthreads();
        GP_STAMP(2);
        const float* const E = reinterpret_cast<const float*>(Z);            // E[2 u + r] = envelope of row r at sample u
#ifdef STOF_GP_SKIP                                              // diagnostic build: transform + envelope only
        continue;
#endif
        for (int r = wave; r < (second ? 2 : 1); r += nwaves) {
            if (r >= 2) break;
            float* const ring = rings + (size_t)(wave & 1) * stof_gp::block_buf_floats(p.cf.radius);
            const stof_gp::EnvPairLds env{E + r};
            if constexpr (MOMENTS) {
                stof_gp::moments_row_blocks(p.cf, tp, ring, lane, env, mom);
            } else {
                float* const out = p.echoes + (row + r) * p.cf.cap * 3;
                RowState st;
                stof_gp::detect_row_blocks(p.cf, tp, ring, lane, env, out, st);      // (word-form blur: the blocked one costs these kernels a wave per SIMD)
                GP_STAMP(3);
                stof_gp::finish_row(st, p.cf, row + r, out, p.reduced, p.counts, p.flags, lane);
                GP_STAMP(4);
            }
        }
    }
    if constexpr (MOMENTS) {
        // one pair of double atomics per work-group: ~90 per microsecond go through on ONE address (2048 groups would
        // queue for 45 us), so the sums are spread over STOF_MOMENT_SLOTS cache lines and folded by stof_gradpeak_fold
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mom[0] += __shfl_xor(mom[0], o);
            mom[1] += __shfl_xor(mom[1], o);
        }
        __syncthreads();                                          // the gradient buffers are free
        double* const red = reinterpret_cast<double*>(tp);       // (the tap image is 16-byte aligned and no longer needed)
        if (lane == 0) { red[wave] = mom[0]; red[4 + wave] = mom[1]; }
        __syncthreads();
        if (tid == 0) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < nwaves; ++w) { a += red[w]; b += red[4 + w]; }
            double* const slot = p.partials + 16 * (blockIdx.x & (STOF_MOMENT_SLOTS - 1));
            atomicAdd(&slot[0], a);
            atomicAdd(&slot[1], b);
        }
    }
}

// stats[0..1] += the slots of a moments launch (one wave; slot s = entries 16 s, 16 s + 1)
__global__ void gradpeak_fold_kernel(const double* __restrict__ partials, double* __restrict__ stats) {
    const int lane = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int s = lane; s < STOF_MOMENT_SLOTS; s += 64) { a += partials[16 * s]; b += partials[16 * s + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (lane == 0) { stats[0] += a; stats[1] += b; }
}

// ----------------------------------------------------------------------------------------------------------------
// toa_detect fused, compile-time FFT plan (row lengths 1536 / 2000 / 2048): the transform of hilbert_ct_kernel
// (fft_small.h analytic_ct: twiddles from a constant table, padded conflict-free slot, rows in registers from the load to
// the un-mixing) in front of the block streamer.  A work-group holds PPW pairs, WPP waves per pair; after the transform
// the envelopes are written back into the slot as a linear interleaved image E[2 u + r] (below the padded positions
// still to be read: position(4 q) <= padded position(4 q)), and wave w of the pair streams row w (WPP = 1: both rows).
// ----------------------------------------------------------------------------------------------------------------
template <int N, int WPP, int PPW, bool MOMENTS>
__global__ __launch_bounds__(64 * WPP * PPW) void toa_fused_ct_kernel(const FusedParams p) {
    using namespace stof_fft;
    static_assert(WPP == 1 || WPP == 2, "one streaming wave per row at most");
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    constexpr int T = 64 * WPP, TT = T * PPW, IO = (N / 4 + T - 1) / T;
    constexpr int TWP = stof_ct::twiddle_lds_entries<N>(), SLOT = ct_slot_entries(N);
    cf* const W = reinterpret_cast<cf*>(lds);
    const int slot = threadIdx.x / T, tid = threadIdx.x % T, lane = tid & 63, wv = tid >> 6;
    cf* const Z = W + TWP + slot * SLOT;
    float* const tp = reinterpret_cast<float*>(W + TWP + PPW * SLOT);            // (TWP and SLOT are even: 16-byte aligned)
    float* const ring = tp + stof_gp::TAPS_LDS + (size_t)(slot * WPP + wv) * stof_gp::block_buf_floats(p.cf.radius);
    const long long npairs = (p.N + 1) / 2, stride = (long long)gridDim.x * PPW;
    long long pr = (long long)blockIdx.x * PPW + slot;
    // (the rows are not kept in registers across the transform as hilbert_ct_kernel does: the streaming stage wants
    // four waves per SIMD, i.e. 128 registers; the un-mixing reads them again -- from L2)
    stof_ct::stage_twiddles<N>(lds, threadIdx.x, TT);
    stof_gp::stage_taps(tp, p.taps, p.cf.radius, threadIdx.x, TT);
    __syncthreads();
    double mom[2] = {0.0, 0.0};
    for (long long p0 = (long long)blockIdx.x * PPW; p0 < npairs; p0 += stride, pr += stride) {
        const bool active = pr < npairs;
        if (WPP == 1 && !active) break;                          // a lone wave: nobody waits for it
        auto sync = [] { if (WPP > 1) __syncthreads(); else wave_lds_sync(); };
        sync();                                                   // previous pair fully streamed
        const float* const xr = p.x + 2 * pr * (size_t)N;
        const float* const x2 = (active && 2 * pr + 1 < p.N) ? xr + N : nullptr;
        if (active) {
            stof_io::PairRegs<IO> cur;
            stof_io::load_pair_regs(cur, xr, x2, N, tid, T);
            stof_io::stage_pair<IO, true>(Z, cur, N, tid, T);
        }
        sync();
        analytic_ct<N, T, CtOpt<true, false>>(Z, W, tid, sync);  // idle slots of a multi-wave group keep the barrier count
        const long long row = 2 * pr;
        const bool second = active && row + 1 < p.N;
        float* const eo = (active && p.env_out) ? p.env_out + row * (size_t)N : nullptr;
        constexpr int nq = N >> 2;
        // analytic signals (padded slot) -> the two envelopes, linear and interleaved, in place.  One wave per pair runs
        // in lockstep, so a piece's reads precede its writes by program order; two waves read everything first.
        auto rows_at = [&](int q, float4& xa4, float4& xb4) {
            xa4 = *reinterpret_cast<const stof_io::f4a*>(xr + 4 * q);
            xb4 = x2 ? *reinterpret_cast<const stof_io::f4a*>(x2 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto emit = [&](int q, const float4& z0, const float4& z1, const float4& xa4, const float4& xb4) {
            const float xa[4] = {xa4.x, xa4.y, xa4.z, xa4.w}, xb[4] = {xb4.x, xb4.y, xb4.z, xb4.w};
            const float re[4] = {z0.x, z0.z, z1.x, z1.z}, im[4] = {z0.y, z0.w, z1.y, z1.w};
            float ea[4], eb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {                          // v1 = Im Z - x2, v2 = x1 - Re Z (pair_io.h)
                ea[e] = stof_io::envelope(xa[e], im[e] - xb[e]);
                eb[e] = stof_io::envelope(xb[e], xa[e] - re[e]);
            }
            stof_io::f4a* d = reinterpret_cast<stof_io::f4a*>(Z + 4 * q);
            d[0] = make_float4(ea[0], eb[0], ea[1], eb[1]);
            d[1] = make_float4(ea[2], eb[2], ea[3], eb[3]);
            if (eo) {
                *reinterpret_cast<float4*>(eo + 4 * q) = make_float4(ea[0], ea[1], ea[2], ea[3]);
                if (second) *reinterpret_cast<float4*>(eo + N + 4 * q) = make_float4(eb[0], eb[1], eb[2], eb[3]);
            }
        };
        if constexpr (WPP == 1) {
            if (active) {
#pragma unroll
                for (int k = 0; k < IO; ++k) {
                    const int q = tid + k * T;
                    if (q < nq) {
                        const stof_io::f4a* sp = reinterpret_cast<const stof_io::f4a*>(Z + 4 * q + 2 * (q >> 2));
                        const float4 z0 = sp[0], z1 = sp[1];
                        float4 xa4, xb4;
                        rows_at(q, xa4, xb4);
                        wave_lds_sync();
                        emit(q, z0, z1, xa4, xb4);
                        wave_lds_sync();
                    }
                }
            }
            wave_lds_sync();
        } else {
            float4 z0[IO], z1[IO], xa4[IO], xb4[IO];
            if (active) {
#pragma unroll
                for (int k = 0; k < IO; ++k) {
                    const int q = tid + k * T;
                    if (q < nq) {
                        const stof_io::f4a* sp = reinterpret_cast<const stof_io::f4a*>(Z + 4 * q + 2 * (q >> 2));
                        z0[k] = sp[0]; z1[k] = sp[1];
                        rows_at(q, xa4[k], xb4[k]);
                    }
                }
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int k = 0; k < IO; ++k) {
                    const int q = tid + k * T;
                    if (q < nq) emit(q, z0[k], z1[k], xa4[k], xb4[k]);
                }
            }
            __syncthreads();
        }
        if (!active) continue;
        const float* const E = reinterpret_cast<const float*>(Z);  // E[2 u + r] = envelope of row r at sample u
        for (int r = wv; r < (second ? 2 : 1); r += WPP) {
            const stof_gp::EnvPairLds env{E + r};
            if constexpr (MOMENTS) {
                stof_gp::moments_row_blocks(p.cf, tp, ring, lane, env, mom);
            } else {
                float* const out = p.echoes + (row + r) * p.cf.cap * 3;
                RowState st;
                stof_gp::detect_row_blocks(p.cf, tp, ring, lane, env, out, st);      // (word-form blur: the blocked one costs these kernels a wave per SIMD)
                stof_gp::finish_row(st, p.cf, row + r, out, p.reduced, p.counts, p.flags, lane);
            }
        }
    }
    if constexpr (MOMENTS) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mom[0] += __shfl_xor(mom[0], o);
            mom[1] += __shfl_xor(mom[1], o);
        }
        __syncthreads();                                          // (every thread leaves the loop: WPP = 1 waves break, others run it out)
        double* const red = reinterpret_cast<double*>(tp);       // the tap image is no longer needed
        const int w = threadIdx.x >> 6;
        if (lane == 0) { red[w] = mom[0]; red[8 + w] = mom[1]; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0, b = 0.0;
            for (int i = 0; i < WPP * PPW; ++i) { a += red[i]; b += red[8 + i]; }
            double* const sl = p.partials + 16 * (blockIdx.x & (STOF_MOMENT_SLOTS - 1));
            atomicAdd(&sl[0], a);
            atomicAdd(&sl[1], b);
        }
    }
}

template <int N, int WPP, int PPW, bool MOMENTS>
int launch_fused_ct(const FusedParams& p, int64_t rows, int radius, hipStream_t stream) {
    const size_t lds = ((size_t)stof_ct::twiddle_lds_entries<N>() + (size_t)PPW * stof_fft::ct_slot_entries(N)) * sizeof(float2) +
                       ((size_t)stof_gp::TAPS_LDS + (size_t)PPW * WPP * stof_gp::block_buf_floats(radius)) * sizeof(float);
    if (lds > (size_t)LDS_BYTES) return STOF_ERR_UNSUPPORTED;
    static stof::LdsLimitOnce once;
    if (int st = once.ensure(reinterpret_cast<const void*>(&toa_fused_ct_kernel<N, WPP, PPW, MOMENTS>), LDS_BYTES)) return st;
    const int64_t npairs = (rows + 1) / 2, groups = (npairs + PPW - 1) / PPW;
    int64_t grid = (int64_t)stof::device_cu_count() * (int64_t)((size_t)LDS_BYTES / lds);
    if (grid > groups) grid = groups;
    hipLaunchKernelGGL((toa_fused_ct_kernel<N, WPP, PPW, MOMENTS>), dim3((unsigned)grid), dim3(64 * WPP * PPW), lds, stream, p);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// row lengths with a compile-time plan; -1: none (or STOF_FUSED_CT=0, or rows that are not 16-byte aligned).  Two waves per
// pair, two pairs per work-group: one wave per pair and four pairs (hilbert_ct_kernel's shape for these lengths) leaves a
// single wave to stream both rows, 80 us on [4096, 2000] against 58.
template <bool MOMENTS>
int try_launch_fused_ct(const FusedParams& p, int64_t rows, int64_t L, int radius, hipStream_t stream) {
    static const int mode = [] { const char* e = getenv("STOF_FUSED_CT"); return e ? atoi(e) : 1; }();   // 0: run-time plan kernel (A/B)
    if (mode == 0 || (reinterpret_cast<size_t>(p.x) & 15) || (p.env_out && (reinterpret_cast<size_t>(p.env_out) & 15))) return -1;
    switch (L) {
        case 1536: return launch_fused_ct<1536, 2, 2, MOMENTS>(p, rows, radius, stream);
        case 2000: return launch_fused_ct<2000, 2, 2, MOMENTS>(p, rows, radius, stream);
        case 2048: return launch_fused_ct<2048, 2, 2, MOMENTS>(p, rows, radius, stream);
        default: return -1;
    }
}

// Beyond this the envelope kernel + row-streaming kernel are faster: with the compile-time-plan Hilbert kernels the two
// launches take 46.8 + 67.0 us on [4096, 4000] against 160 us fused (run-time plan transform inside), and 25.9 + 35.1 us
// against 64 us on [4096, 2000], where the single launch is kept for its lower host overhead and no envelope buffer.
constexpr int FUSED_MAX_L = 2048;

size_t fused_lds_bytes(int64_t n, int radius, stof_fft::Plan* plan) {
    static const int64_t max_l = [] { const char* e = getenv("STOF_FUSED_MAX_L"); return e ? (int64_t)atoll(e) : (int64_t)FUSED_MAX_L; }();
    if (n > max_l || !stof::hilbert_fast_lds_bytes(n, plan)) return 0;
    return ((size_t)n + stof_fft::twiddle_entries((int)n) + 1) * sizeof(float2) +
           ((size_t)stof_gp::TAPS_LDS + 2 * stof_gp::block_buf_floats(radius)) * sizeof(float);
}

bool bad_common(int64_t N, int64_t L, int32_t grad_step, int32_t radius, int64_t cap) {
    return N < 0 || L < 0 || radius < 0 || cap < 0 || grad_step <= 0;        // rescale_factor < 6: the reference fails too
}

Config make_config(int64_t L, int32_t grad_step, int32_t radius, float threshold, int32_t ival_min, int32_t ival_max,
                   int64_t cap, int64_t echo_max) {
    Config cf;
    cf.L = (int)L; cf.spacing = (float)grad_step; cf.radius = radius;
    cf.th_pos = threshold; cf.th_neg = -threshold / 4.0f;             // models/gradpeak.py:19
    // the gate ival_min < am - ap < ival_max is evaluated in 32-bit arithmetic on positions below 2^30
    cf.ival_min = ival_min < -(1 << 30) ? -(1 << 30) : ival_min;
    cf.ival_max = ival_max > (1 << 30) ? (1 << 30) : ival_max;
    cf.cap = cap; cf.echo_max = echo_max > 0 ? echo_max : 0;
    return cf;
}

// launch of gradpeak_rows_kernel with its gradient buffers
template <bool MOMENTS, int WAVES>
int launch_rows(const float* env, int64_t N, const Config& cf, const float* taps, const float* th_dev, float* echoes, float* reduced,
                int* counts, int* flags, double* stats, float* blurred, int64_t groups_per_cu, hipStream_t stream) {
    int ring_stride = stof_gp::block_buf_floats(cf.radius);
    if (blurred && stof_gp::ring_floats(cf.radius) > ring_stride) ring_stride = stof_gp::ring_floats(cf.radius);   // stream_words' ring
    const size_t dyn = (size_t)WAVES * ring_stride * sizeof(float);
    int64_t grid = (N + WAVES - 1) / WAVES;
    const int64_t maxg = (int64_t)stof::device_cu_count() * groups_per_cu;
    if (grid > maxg) grid = maxg;
    hipLaunchKernelGGL((gradpeak_rows_kernel<MOMENTS, WAVES>), dim3((unsigned)grid), dim3(64 * WAVES), dyn, stream, env, (long long)N, cf,
                       taps, th_dev, echoes, reduced, counts, flags, stats, blurred, ring_stride);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

}  // namespace

extern "C" int stof_gradpeak_moments(const float* env, int64_t N, int64_t L, int32_t grad_step, const float* taps,
                                     int32_t radius, double* stats, void* stream) {
    if (!env || !taps || !stats || bad_common(N, L, grad_step, radius, 0)) return STOF_ERR_BAD_ARG;
    if (N == 0 || L == 0) return STOF_OK;
    if (radius > stof_gp::MAXRAD || L > 0x3fffffffLL) return STOF_ERR_UNSUPPORTED;
    const Config cf = make_config(L, grad_step, radius, 0.f, 0, 0, 0, 0);
    {
        int st = STOF_OK;
        if (launch_split<true>(env, N, cf, taps, nullptr, nullptr, nullptr, nullptr, nullptr, stats, static_cast<hipStream_t>(stream), &st))
            return st;
    }
    return launch_rows<true, MOMENT_WAVES>(env, N, cf, taps, nullptr, nullptr, nullptr, nullptr, nullptr, stats, nullptr, 2,
                                           static_cast<hipStream_t>(stream));
}

extern "C" int64_t stof_gradpeak_blurred_stride(int64_t L, int32_t radius) {
    return (L <= 0 || radius < 0) ? 0 : (int64_t)stof_gp::word_count((int)L, radius) * 64;
}

extern "C" int stof_gradpeak_moments_store(const float* env, int64_t N, int64_t L, int32_t grad_step, const float* taps,
                                           int32_t radius, double* stats, float* blurred, void* stream) {
    if (!env || !taps || !stats || !blurred || bad_common(N, L, grad_step, radius, 0)) return STOF_ERR_BAD_ARG;
    if (N == 0 || L == 0) return STOF_OK;
    if (radius > stof_gp::MAXRAD || L > 0x3fffffffLL) return STOF_ERR_UNSUPPORTED;
    const Config cf = make_config(L, grad_step, radius, 0.f, 0, 0, 0, 0);
    return launch_rows<true, MOMENT_WAVES>(env, N, cf, taps, nullptr, nullptr, nullptr, nullptr, nullptr, stats, blurred, 2,
                                           static_cast<hipStream_t>(stream));
}

extern "C" int stof_gradpeak_threshold(const double* stats, float* threshold_out, void* stream) {
    if (!stats || !threshold_out) return STOF_ERR_BAD_ARG;
    hipLaunchKernelGGL(gradpeak_threshold_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), stats, threshold_out);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_grad_peak_detect(const float* env, int64_t N, int64_t L, int32_t grad_step, const float* taps,
                                     int32_t radius, float threshold, const float* threshold_dev, int32_t ival_min,
                                     int32_t ival_max, int64_t echo_max, float* echoes, int64_t cap, float* reduced,
                                     int32_t* counts, int32_t* flags, void* stream) {
    if (!env || !taps || !counts || !flags || (!echoes && cap > 0) || bad_common(N, L, grad_step, radius, cap))
        return STOF_ERR_BAD_ARG;
    if (echo_max > 0 && !reduced) return STOF_ERR_BAD_ARG;
    if (echo_max > 0 && cap > 4096) return STOF_ERR_UNSUPPORTED;      // reduce_row selects among <= 64 x 64 entries per row
    if (N == 0) return STOF_OK;
    if (radius > stof_gp::MAXRAD || L > 0x3fffffffLL || N > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    const Config cf = make_config(L, grad_step, radius, threshold, ival_min, ival_max, cap, echo_max);
    if (hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), static_cast<hipStream_t>(stream)) != hipSuccess) return STOF_ERR_HIP;
    {
        int st = STOF_OK;
        if (launch_split<false>(env, N, cf, taps, threshold_dev, echoes, reduced, counts, flags, nullptr, static_cast<hipStream_t>(stream), &st))
            return st;
    }
    return launch_rows<false, ROWS_WAVES>(env, N, cf, taps, threshold_dev, echoes, reduced, counts, flags, nullptr, nullptr, 8,
                                          static_cast<hipStream_t>(stream));
}

extern "C" int stof_grad_peak_detect_blurred(const float* env, const float* blurred, int64_t N, int64_t L, int32_t radius,
                                             float threshold, const float* threshold_dev, int32_t ival_min, int32_t ival_max,
                                             int64_t echo_max, float* echoes, int64_t cap, float* reduced, int32_t* counts,
                                             int32_t* flags, void* stream) {
    if (!env || !blurred || !counts || !flags || (!echoes && cap > 0) || bad_common(N, L, 1, radius, cap)) return STOF_ERR_BAD_ARG;
    if (echo_max > 0 && !reduced) return STOF_ERR_BAD_ARG;
    if (echo_max > 0 && cap > 4096) return STOF_ERR_UNSUPPORTED;      // reduce_row selects among <= 64 x 64 entries per row
    if (N == 0) return STOF_OK;
    if (radius > stof_gp::MAXRAD || L > 0x3fffffffLL || N > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    const Config cf = make_config(L, 1, radius, threshold, ival_min, ival_max, cap, echo_max);
    if (hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), static_cast<hipStream_t>(stream)) != hipSuccess) return STOF_ERR_HIP;
    int64_t grid = (N + ROWS_WAVES - 1) / ROWS_WAVES;
    const int64_t maxg = (int64_t)stof::device_cu_count() * 8;
    if (grid > maxg) grid = maxg;
    hipLaunchKernelGGL(gradpeak_flags_kernel, dim3((unsigned)grid), dim3(64 * ROWS_WAVES), 0, static_cast<hipStream_t>(stream), env,
                       blurred, (long long)N, cf, threshold_dev, echoes, reduced, counts, flags);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_toa_detect_fused_ok(int64_t L, int32_t radius) {
    return fused_lds_bytes(L, radius, nullptr) != 0 && radius <= stof_gp::MAXRAD;
}

namespace {

// shared launch of the fused kernel (detection or moments pre-pass)
template <bool MOMENTS>
int launch_fused(FusedParams& p, int64_t N, int64_t L, int32_t radius, hipStream_t stream) {
    {
        const int st = try_launch_fused_ct<MOMENTS>(p, N, L, radius, stream);
        if (st != -1 && st != STOF_ERR_UNSUPPORTED) return st;
    }
    const size_t lds = fused_lds_bytes(L, radius, &p.plan);
    if (!lds) return STOF_ERR_UNSUPPORTED;            // caller falls back to stof_hilbert + stof_grad_peak_detect
    static stof::LdsLimitOnce once;
    if (int st = once.ensure(reinterpret_cast<const void*>(&toa_fused_kernel<MOMENTS>), LDS_BYTES)) return st;
    const int64_t npairs = (N + 1) / 2;
    int64_t per_cu = (int64_t)LDS_BYTES / (int64_t)lds;
    if (per_cu > 16) per_cu = 16;
    int threads = 64;                                              // 16 waves per CU whatever the row length
    while (threads < 256 && per_cu * (threads / 64) < 16) threads *= 2;
    if (const char* e = getenv("STOF_FUSED_THREADS")) threads = atoi(e);       // tuning / A-B switch
    int64_t grid = (int64_t)stof::device_cu_count() * per_cu;
    if (grid > npairs) grid = npairs;
    hipLaunchKernelGGL(toa_fused_kernel<MOMENTS>, dim3((unsigned)grid), dim3(threads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

}  // namespace

extern "C" int stof_toa_detect(const float* frame, int64_t N, int64_t L, int32_t grad_step, const float* taps, int32_t radius,
                               float threshold, int32_t ival_min, int32_t ival_max, int64_t echo_max, float* echoes,
                               int64_t cap, float* reduced, int32_t* counts, int32_t* flags, float* env_out, void* stream) {
    if (!frame || !taps || !counts || !flags || (!echoes && cap > 0) || bad_common(N, L, grad_step, radius, cap))
        return STOF_ERR_BAD_ARG;
    if (echo_max > 0 && !reduced) return STOF_ERR_BAD_ARG;
    if (echo_max > 0 && cap > 4096) return STOF_ERR_UNSUPPORTED;      // reduce_row selects among <= 64 x 64 entries per row
    if (N == 0) return STOF_OK;
    if (N > 0x7fffffffLL || radius > stof_gp::MAXRAD) return STOF_ERR_UNSUPPORTED;
    if (!fused_lds_bytes(L, radius, nullptr)) return STOF_ERR_UNSUPPORTED;
    FusedParams p;
    p.x = frame; p.taps = taps; p.echoes = echoes; p.reduced = echo_max > 0 ? reduced : nullptr;
    p.counts = counts; p.flags = flags; p.env_out = env_out; p.partials = nullptr; p.N = N;
    p.cf = make_config(L, grad_step, radius, threshold, ival_min, ival_max, cap, echo_max);
    if (hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), static_cast<hipStream_t>(stream)) != hipSuccess) return STOF_ERR_HIP;
    return launch_fused<false>(p, N, L, radius, static_cast<hipStream_t>(stream));
}

#ifdef STOF_GP_STAMPS
extern "C" int stof_debug_gp_stamps(unsigned long long* out, int reset) {      // out[8]: sums over the slots
    static unsigned long long host[GP_STAMP_SLOTS][8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gp_stamps), sizeof(host)) != hipSuccess) return STOF_ERR_HIP;
    for (int i = 0; i < 8; ++i) { out[i] = 0; for (int sl = 0; sl < GP_STAMP_SLOTS; ++sl) out[i] += host[sl][i]; }
    if (reset) { memset(host, 0, sizeof(host)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_gp_stamps), host, sizeof(host)) != hipSuccess) return STOF_ERR_HIP; }
    return STOF_OK;
}
#endif

extern "C" int stof_toa_moments(const float* frame, int64_t N, int64_t L, int32_t grad_step, const float* taps, int32_t radius,
                                float* env_out, double* partials, double* stats, void* stream) {
    if (!frame || !taps || !env_out || !partials || !stats || bad_common(N, L, grad_step, radius, 0)) return STOF_ERR_BAD_ARG;
    if (N == 0 || L == 0) return STOF_OK;
    if (N > 0x7fffffffLL || radius > stof_gp::MAXRAD) return STOF_ERR_UNSUPPORTED;
    if (!fused_lds_bytes(L, radius, nullptr)) return STOF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(partials, 0, (size_t)STOF_MOMENT_SLOTS * 16 * sizeof(double), s) != hipSuccess) return STOF_ERR_HIP;
    FusedParams p;
    p.x = frame; p.taps = taps; p.echoes = nullptr; p.reduced = nullptr; p.counts = nullptr; p.flags = nullptr;
    p.env_out = env_out; p.partials = partials; p.N = N;
    p.cf = make_config(L, grad_step, radius, 0.f, 0, 0, 0, 0);
    if (int st = launch_fused<true>(p, N, L, radius, s)) return st;
    hipLaunchKernelGGL(gradpeak_fold_kernel, dim3(1), dim3(64), 0, s, partials, stats);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
