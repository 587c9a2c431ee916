// grad_peak_detect (models/gradpeak.py:8-68) in two gfx950 kernels.
//
//   gradpeak_gradient : torch.gradient(env, spacing=g) (:14) -> Gaussian blur (:15, :89-96),
//                       zero padded; also the batch-wide sum / sum of squares (double) that
//                       the default threshold needs (:18, Q7).
//   gradpeak_pair     : one wavefront per row walks the row 64 samples at a time and does the
//                       whole hysteresis pairing (:23-60) with ballots: rising edges of
//                       grad > th (onset candidates `ap`) and of grad < -th/4 (peaks `am`),
//                       nearest preceding ap per am, interval gate, first am per distinct ap.
#include <hip/hip_runtime.h>
#include "stof_common.h"

namespace {

constexpr int GP_CH = 2048;        // output samples per chunk in the gradient kernel
constexpr int GP_MAXRAD = 96;

__global__ __launch_bounds__(256) void gradpeak_gradient_kernel(const float* __restrict__ env, int L, float spacing,
                                                                const float* __restrict__ taps, int radius,
                                                                float* __restrict__ grad,
                                                                double* __restrict__ stats) {
    __shared__ float e[GP_CH + 2 * GP_MAXRAD + 2];
    __shared__ float tp[2 * GP_MAXRAD + 1];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row = blockIdx.x;
    const float* er = env + row * (size_t)L;
    float* gr = grad + row * (size_t)L;
    const int ntaps = 2 * radius + 1;
    for (int i = tid; i < ntaps; i += 256) tp[i] = taps[i];
    const float two_sp = 2.0f * spacing;
    double s1 = 0.0, s2 = 0.0;
    for (int c0 = 0; c0 < L; c0 += GP_CH) {
        __syncthreads();
        // e[i] <-> env[c0 - radius - 1 + i]
        for (int i = tid; i < GP_CH + 2 * radius + 2; i += 256) {
            const int t = c0 - radius - 1 + i;
            e[i] = (t >= 0 && t < L) ? er[t] : 0.f;
        }
        __syncthreads();
        for (int o = tid; o < GP_CH; o += 256) {
            const int t = c0 + o;
            if (t >= L) break;
            float acc = 0.f;
            for (int j = 0; j < ntaps; ++j) {
                const int u = t + j - radius;            // gradient sample index
                float g = 0.f;                           // zero padding of the blur (:94)
                if (u >= 0 && u < L) {
                    const int b = o + j + 1;             // e index of env[u]
                    if (L == 1) g = 0.f;
                    else if (u == 0) g = (e[b + 1] - e[b]) / spacing;
                    else if (u == L - 1) g = (e[b] - e[b - 1]) / spacing;
                    else g = (e[b + 1] - e[b - 1]) / two_sp;
                }
                acc = fmaf(tp[j], g, acc);
            }
            gr[t] = acc;
            s1 += (double)acc;
            s2 += (double)acc * (double)acc;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (tid == 0 && stats) {
        atomicAdd(&stats[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&stats[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void gradpeak_pair_kernel(const float* __restrict__ env,
                                                            const float* __restrict__ grad, int N, int L,
                                                            float th_pos, float th_neg, int ival_min, int ival_max,
                                                            float* __restrict__ echoes, long long cap,
                                                            int* __restrict__ counts, int* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= N) return;                                   // whole wave exits together
    const float* g = grad + row * (long long)L;
    const float* e = env + row * (long long)L;
    float* out = echoes + row * cap * 3;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long le_mask = lt_mask | (1ull << lane);

    int last_ap = -1;          // most recent rising-slope edge seen so far (carry across chunks)
    int last_kept_ap = -1;     // onset of the last surviving candidate (uniqueness, :58-59)
    int nout = 0;
    int any_ap = 0, any_am = 0;
    for (int c0 = 0; c0 < L - 1; c0 += 64) {
        const int i = c0 + lane;
        bool ep = false, em = false;
        if (i < L - 1) {
            const float a = g[i], b = g[i + 1];
            ep = !(a > th_pos) && (b > th_pos);             // diff(int(grad > th)) == 1  (:27,:29)
            em = !(a < th_neg) && (b < th_neg);             // diff(int(grad < -th/4)) == 1 (:28,:30)
        }
        const unsigned long long pm = __ballot(ep);
        const unsigned long long mm = __ballot(em);
        any_ap |= (pm != 0);
        any_am |= (mm != 0);
        // nearest preceding (<=) onset candidate for this lane's peak candidate (:42-45)
        const unsigned long long below = pm & le_mask;
        const int ap = below ? (c0 + 63 - __builtin_clzll(below)) : last_ap;
        const int gap = i - ap;
        const bool valid = em && (ap >= 0) && (gap > ival_min) && (gap < ival_max);   // :48-49
        const unsigned long long vm = __ballot(valid);
        // onset of the previous surviving candidate (previous valid lane, else the carry)
        const unsigned long long vbelow = vm & lt_mask;
        const int prev_lane = vbelow ? (63 - __builtin_clzll(vbelow)) : 0;
        const int prev_ap_lane = __shfl(ap, prev_lane);
        const int prev_ap = vbelow ? prev_ap_lane : last_kept_ap;
        const bool keep = valid && (ap != prev_ap);          // first am per distinct ap (:58-59)
        const unsigned long long km = __ballot(keep);
        if (keep) {
            const int pos = nout + __builtin_popcountll(km & lt_mask);
            if (pos < cap) {
                out[3 * pos + 0] = (float)ap;
                out[3 * pos + 1] = (float)i;
                out[3 * pos + 2] = e[i];                     // data[i, am] (:66)
            }
        }
        nout += __builtin_popcountll(km);
        if (vm) last_kept_ap = __shfl(ap, 63 - __builtin_clzll(vm));
        if (pm) last_ap = c0 + 63 - __builtin_clzll(pm);
    }
    if (lane == 0) {
        counts[row] = nout;
        if (any_ap && any_am && nout == 0) atomicOr(&flags[0], 1);   // Q9 (:54-55)
        if (nout > 0) atomicMax(&flags[1], nout);                    // Kmax of the batch (one host read for both)
    }
}

}  // namespace

extern "C" int stof_gradpeak_gradient(const float* env, int64_t N, int64_t L, int32_t grad_step,
                                      const float* taps, int32_t radius, float* grad, double* stats,
                                      void* stream) {
    if (!env || !taps || !grad || N < 0 || L < 0 || radius < 0) return STOF_ERR_BAD_ARG;
    if (grad_step <= 0) return STOF_ERR_BAD_ARG;       // rescale_factor < 6: the reference fails too
    if (N == 0 || L == 0) return STOF_OK;
    if (radius > GP_MAXRAD || N > 0x7fffffffLL || L > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gradpeak_gradient_kernel, dim3((unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       env, (int)L, (float)grad_step, taps, (int)radius, grad, stats);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_gradpeak_pair(const float* env, const float* grad, int64_t N, int64_t L, float thres_pos,
                                  int32_t ival_min, int32_t ival_max, float* echoes, int64_t cap,
                                  int32_t* counts, int32_t* flags, void* stream) {
    if (!env || !grad || !counts || !flags || (!echoes && cap > 0) || N < 0 || L < 0 || cap < 0)
        return STOF_ERR_BAD_ARG;
    if (N == 0) return STOF_OK;
    if (N > 0x7fffffffLL || L > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    const float th_neg = -thres_pos / 4.0f;              // models/gradpeak.py:19
    hipLaunchKernelGGL(gradpeak_pair_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), env, grad, (int)N, (int)L, thres_pos, th_neg,
                       (int)ival_min, (int)ival_max, echoes, (long long)cap, counts, flags);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
