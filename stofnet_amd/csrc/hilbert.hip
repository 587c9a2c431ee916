// hilbert_transform (utils/hilbert.py:5-21): v = ifft(H .* fft(y)) along the last dim with
//   H = [1, 2 (bins 1..n/2-1), 1 (bin n/2), 0 ...]            (Q6: for odd n bin n/2 is not doubled)
//
// Three kernels:
//   hilbert_ct_kernel      (row lengths 1536 / 2000 / 2048 / 4000 / 4096: compile-time plan, see below)
//   hilbert_pairs_kernel   (fast path: n even with prime factors 2, 3, 5 only and 8 n bytes within the LDS budget)
//       one work-group per PAIR of rows; the pair rides one complex transform z = x1 + i x2 that lives in LDS; few
//       passes of register butterflies of radix up to 25 with the filter fused into the middle pass (fft_small.h);
//       twiddles from a two-level table the work-group builds in double precision.
//   hilbert_generic_kernel (every other length: odd n, other prime factors, rows beyond LDS)
//       stage-by-stage mixed radix (4, 2, 3, 5, then any prime) as described below; for rows that do not fit LDS the same
//       stages run on a per-work-group scratch in global memory (L2 / Infinity Cache resident), so any n is served.
//
// hilbert_generic_kernel: one work-group per row (pair); the row lives in LDS as complex fp32 for its whole life:
//   forward  : in-place decimation-in-frequency, mixed radix (4, 2, 3, 5, then any prime)
//              -> spectrum in digit-reversed positions
//   filter   : H[k]/n applied at each position's true frequency k
//   inverse  : in-place decimation-in-time with the radices in reverse order, consuming the
//              digit-reversed spectrum -> natural order, so no permutation pass exists.
// Twiddles exp(-2 pi i k/n) come from a per-call table in the workspace (double-precision
// sincospi, rounded once).  Prime radices > 5 run as out-of-place O(R) sums per output
// through a second LDS buffer.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "stof_common.h"
#include "stof_hip_util.h"
#include "fft_small.h"
#include "pair_io.h"
#include "ct_twiddles.h"

namespace {

constexpr int MAX_STAGES = 24;
constexpr int LDS_BYTES = 160 * 1024;
constexpr int GENERIC_ZG_GRID = 512;     // work-groups of the global-scratch mode (2 per CU)

struct FftPlan {
    int n;
    int nstages;
    int radix[MAX_STAGES];
    int needs_second;     // a prime radix > 5 is present
};

__host__ void make_plan(int n, FftPlan* p) {
    p->n = n;
    p->nstages = 0;
    p->needs_second = 0;
    int m = n;
    while (m % 4 == 0) { p->radix[p->nstages++] = 4; m /= 4; }
    while (m % 2 == 0) { p->radix[p->nstages++] = 2; m /= 2; }
    while (m % 3 == 0) { p->radix[p->nstages++] = 3; m /= 3; }
    while (m % 5 == 0) { p->radix[p->nstages++] = 5; m /= 5; }
    for (int f = 7; (long long)f * f <= m; f += 2)
        while (m % f == 0) { p->radix[p->nstages++] = f; m /= f; p->needs_second = 1; }
    if (m > 1) { p->radix[p->nstages++] = m; p->needs_second = 1; }
}

// Per-call tables in the workspace: tw[k] = exp(-2 pi i k/n) (double-precision sincospi, rounded
// once) and hf[p] = H[k(p)]/n, the Hilbert filter (utils/hilbert.py:13-17) at the true frequency
// k(p) of digit-reversed position p, with the 1/n of the inverse transform folded in.
__global__ void tables_kernel(float2* __restrict__ tw, float* __restrict__ hf, const FftPlan plan) {
    const int n = plan.n;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double s, c;
    sincospi(-2.0 * (double)p / (double)n, &s, &c);
    tw[p] = make_float2((float)c, (float)s);
    int rem = p, mm = n, k = 0, mult = 1;
    for (int st = 0; st < plan.nstages; ++st) {
        const int R = plan.radix[st];
        mm /= R;
        const int q = rem / mm;
        rem -= q * mm;
        k += q * mult;
        mult *= R;
    }
    const int nyq = n / 2;
    const float h = (k == 0 || k == nyq) ? 1.f : (k < nyq ? 2.f : 0.f);
    hf[p] = h / (float)n;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {   // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// y_q = sum_k x_k w^{qk}, w = exp(-+2 pi i/R); INV selects the conjugate root.
template <int R, bool INV>
__device__ __forceinline__ void small_dft(float2 (&x)[R], const float2* __restrict__ tw, int n) {
    if constexpr (R == 2) {
        const float2 a = x[0], b = x[1];
        x[0] = cadd(a, b);
        x[1] = csub(a, b);
    } else if constexpr (R == 4) {
        const float2 s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]);
        const float2 s13 = cadd(x[1], x[3]), d13 = csub(x[1], x[3]);
        // forward: w = -i  -> (-i) * d13 = (d13.y, -d13.x); inverse: w = +i -> (-d13.y, d13.x)
        const float2 rot = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
        x[0] = cadd(s02, s13);
        x[1] = cadd(d02, rot);
        x[2] = csub(s02, s13);
        x[3] = csub(d02, rot);
    } else if constexpr (R == 3) {
        // w = exp(-+2 pi i/3) = -1/2 -+ i sqrt(3)/2
        const float2 s12 = cadd(x[1], x[2]), d12 = csub(x[1], x[2]);
        const float2 t = make_float2(fmaf(-0.5f, s12.x, x[0].x), fmaf(-0.5f, s12.y, x[0].y));
        const float sn = INV ? 0.8660254037844386f : -0.8660254037844386f;
        const float2 rot = make_float2(-sn * d12.y, sn * d12.x);          // i * sn * d12
        x[0] = cadd(x[0], s12);
        x[1] = cadd(t, rot);
        x[2] = csub(t, rot);
    } else {
        static_assert(R == 5, "small_dft handles radix 2, 3, 4, 5");
        // Winograd-style radix 5: c1 = cos(2pi/5), c2 = cos(4pi/5), s1 = sin(2pi/5), s2 = sin(4pi/5)
        constexpr float c1 = 0.30901699437494745f, c2 = -0.8090169943749475f;
        const float s1 = INV ? 0.9510565162951535f : -0.9510565162951535f;
        const float s2 = INV ? 0.5877852522924731f : -0.5877852522924731f;
        const float2 a14 = cadd(x[1], x[4]), b14 = csub(x[1], x[4]);
        const float2 a23 = cadd(x[2], x[3]), b23 = csub(x[2], x[3]);
        const float2 t1 = make_float2(x[0].x + c1 * a14.x + c2 * a23.x, x[0].y + c1 * a14.y + c2 * a23.y);
        const float2 t2 = make_float2(x[0].x + c2 * a14.x + c1 * a23.x, x[0].y + c2 * a14.y + c1 * a23.y);
        // i * (s1 b14 + s2 b23) and i * (s2 b14 - s1 b23)
        const float2 u1 = make_float2(-(s1 * b14.y + s2 * b23.y), s1 * b14.x + s2 * b23.x);
        const float2 u2 = make_float2(-(s2 * b14.y - s1 * b23.y), s2 * b14.x - s1 * b23.x);
        x[0] = cadd(x[0], cadd(a14, a23));
        x[1] = cadd(t1, u1);
        x[4] = csub(t1, u1);
        x[2] = cadd(t2, u2);
        x[3] = csub(t2, u2);
    }
}

// One in-place stage over the whole row.  m = current block length, sub = m / R.
template <int R, bool INV>
__device__ void stage_small(float2* __restrict__ d, const float2* __restrict__ tw, int n, int m) {
    const int sub = m / R;
    const int tstep = n / m;
    const float inv_sub = 1.0f / (float)sub;
    for (int b = threadIdx.x; b < n / R; b += blockDim.x) {
        // b / sub without integer division: exact for b < 2^22 (float quotient + one correction)
        int blk = (int)((float)b * inv_sub);
        int j = b - blk * sub;
        if (j < 0) { j += sub; blk -= 1; } else if (j >= sub) { j -= sub; blk += 1; }
        float2* base = d + blk * m + j;
        float2 x[R];
#pragma unroll
        for (int k = 0; k < R; ++k) x[k] = base[k * sub];
        if (INV) {
#pragma unroll
            for (int q = 1; q < R; ++q) x[q] = cmulc(x[q], tw[(size_t)j * q * tstep]);
        }
        small_dft<R, INV>(x, tw, n);
        if (!INV) {
#pragma unroll
            for (int q = 1; q < R; ++q) x[q] = cmul(x[q], tw[(size_t)j * q * tstep]);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) base[k * sub] = x[k];
    }
}

// Prime radix R > 5: out-of-place, one output element per thread iteration.
template <bool INV>
__device__ void stage_generic(const float2* __restrict__ src, float2* __restrict__ dst,
                              const float2* __restrict__ tw, int n, int m, int R) {
    const int sub = m / R;
    const int tstep = n / m;
    const int rstep = n / R;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int blk = e / m, rem = e - blk * m;
        const int q = rem / sub, j = rem - q * sub;
        const float2* base = src + blk * m + j;
        float2 acc = make_float2(0.f, 0.f);
        int idx = 0;                                  // (q*k) mod R
        for (int k = 0; k < R; ++k) {
            float2 v = base[k * sub];
            if (INV) v = cmulc(v, tw[(size_t)j * k * tstep]);      // input twiddle (k plays q's role)
            const float2 w = tw[(size_t)idx * rstep];
            acc = cadd(acc, INV ? cmulc(v, w) : cmul(v, w));
            idx += q;
            if (idx >= R) idx -= R;
        }
        if (!INV) acc = cmul(acc, tw[(size_t)j * q * tstep]);
        dst[e] = acc;
    }
}

template <bool INV>
__device__ void run_stage(float2*& cur, float2*& other, const float2* tw, int n, int m, int R) {
    switch (R) {
        case 2: stage_small<2, INV>(cur, tw, n, m); break;
        case 3: stage_small<3, INV>(cur, tw, n, m); break;
        case 4: stage_small<4, INV>(cur, tw, n, m); break;
        case 5: stage_small<5, INV>(cur, tw, n, m); break;
        default: {
            stage_generic<INV>(cur, other, tw, n, m, R);
            float2* t = cur; cur = other; other = t;
        }
    }
    __syncthreads();
}

// One work-group transforms rows blockIdx.x, blockIdx.x + gridDim.x, ... so the twiddle table is
// brought into LDS once per work-group (TW_LDS) instead of being fetched from L2 in every stage.
//
// PAIR (even n): two real rows ride one complex transform, z = x1 + i x2.  The filter is linear, so
// ifft(H fft(z)) = a1 + i a2 with a_j = x_j + i v_j the analytic signals (for even n the filter
// returns the real part unchanged), i.e. Re = x1 - v2, Im = v1 + x2, hence v1 = Im - x2 and
// v2 = x1 - Re: half the transforms per row.  For odd n (Q6: bin n/2 not doubled) the real part is
// not preserved and every row gets its own transform.
// ZG: the row does not fit LDS; buf0 / buf1 are this work-group's slice of `scratch` (global memory).  The stages are
// separated by __syncthreads(), whose work-group-scope release/acquire also orders the global accesses of one
// work-group (all its waves share the CU's L1).
template <bool TW_LDS, bool PAIR, bool ZG>
__global__ __launch_bounds__(512) void hilbert_generic_kernel(const float* __restrict__ x, const float2* __restrict__ twg,
                                                              const float* __restrict__ hf, const FftPlan plan,
                                                              long long nrows, float* __restrict__ env,
                                                              float* __restrict__ re, float* __restrict__ im,
                                                              float2* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int n = plan.n;
    float2* const buf0 = ZG ? scratch + (size_t)blockIdx.x * 2 * (size_t)n : lds;
    float2* const buf1 = buf0 + n;                                 // only used with a prime radix > 5
    const float2* tw = twg;
    if (TW_LDS) {
        float2* twl = lds + (plan.needs_second ? 2 : 1) * (size_t)n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) twl[i] = twg[i];
        tw = twl;
    }
    constexpr int RPW = PAIR ? 2 : 1;                              // rows per transform
    for (long long row = (long long)blockIdx.x * RPW; row < nrows; row += (long long)gridDim.x * RPW) {
        float2* cur = buf0;
        float2* other = buf1;
        const float* xr = x + row * (size_t)n;
        const bool second = PAIR && row + 1 < nrows;
        __syncthreads();                                           // previous row fully written out
        for (int i = threadIdx.x; i < n; i += blockDim.x) cur[i] = make_float2(xr[i], second ? xr[n + i] : 0.f);
        __syncthreads();

        // forward DIF: block length shrinks n -> 1
        int m = n;
        for (int s = 0; s < plan.nstages; ++s) {
            run_stage<false>(cur, other, tw, n, m, plan.radix[s]);
            m /= plan.radix[s];
        }
        // filter: H[k(p)]/n per digit-reversed position, from the per-call table
        for (int p = threadIdx.x; p < n; p += blockDim.x) {
            const float sc = hf[p];
            const float2 v = cur[p];
            cur[p] = make_float2(v.x * sc, v.y * sc);
        }
        __syncthreads();
        // inverse DIT: stages in reverse order, block length grows 1 -> n
        for (int s = plan.nstages - 1; s >= 0; --s) {
            m *= plan.radix[s];
            run_stage<true>(cur, other, tw, n, m, plan.radix[s]);
        }
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float2 v = cur[i];
            if (!PAIR) {
                if (env) env[row * (size_t)n + i] = hypotf(v.x, v.y);        // torch.abs of a complex64
                if (re) re[row * (size_t)n + i] = v.x;
                if (im) im[row * (size_t)n + i] = v.y;
            } else {
                const float x1 = xr[i], x2 = second ? xr[n + i] : 0.f;
                const float v1 = v.y - x2, v2 = x1 - v.x;
                if (env) env[row * (size_t)n + i] = hypotf(x1, v1);
                if (re) re[row * (size_t)n + i] = x1;
                if (im) im[row * (size_t)n + i] = v1;
                if (second) {
                    if (env) env[(row + 1) * (size_t)n + i] = hypotf(x2, v2);
                    if (re) re[(row + 1) * (size_t)n + i] = x2;
                    if (im) im[(row + 1) * (size_t)n + i] = v2;
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Fast path: one work-group per pair of rows, the pair's complex transform lives in LDS (fft_small.h).
// PAIR un-mixing (even n): z = x1 + i x2, ifft(H fft(z)) = a1 + i a2 with a_j = x_j + i v_j the analytic signals,
// so Re = x1 - v2, Im = v1 + x2, hence v1 = Im - x2, v2 = x1 - Re.
// ----------------------------------------------------------------------------------------------------------------
template <bool NT>      // NT: envelope written with non-temporal stores (stof_hilbert_streamed)
__global__ __launch_bounds__(512) void hilbert_pairs_kernel(const float* __restrict__ x, const stof_fft::Plan plan,
                                                            long long nrows, float* __restrict__ env,
                                                            float* __restrict__ re, float* __restrict__ im) {
    using namespace stof_fft;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int n = plan.n, tid = threadIdx.x, T = blockDim.x;
    cf* const Z = reinterpret_cast<cf*>(lds);
    cf* const ta = Z + n;
    cf* const tb = ta + TW_A;
    const int nb = (n + TW_A - 1) / TW_A;
    for (int t = tid; t < TW_A + nb; t += T) {                   // twiddle tables: double-precision sincospi, rounded once
        const double k = t < TW_A ? (double)t : (double)(t - TW_A) * (double)TW_A;
        double sn, cs;
        sincospi(-2.0 * k / (double)n, &sn, &cs);
        ta[t] = mk((float)cs, (float)sn);                        // tb follows ta in LDS
    }
    const Twiddles tw{ta, tb};
    const long long npairs = (nrows + 1) / 2;
    for (long long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const long long row = 2 * pr;
        const float* xr = x + row * (size_t)n;
        const bool second = row + 1 < nrows;
        __syncthreads();                                          // tables built / previous pair fully written out
        const float* const x2 = second ? xr + n : nullptr;
        stof_io::load_pair(Z, xr, x2, n, tid, T);
        __syncthreads();
        analytic_in_place(Z, plan, tw, tid, T, [] { __syncthreads(); });
        float* const e1 = env ? env + row * (size_t)n : nullptr;
        float* const r1 = re ? re + row * (size_t)n : nullptr;
        float* const i1 = im ? im + row * (size_t)n : nullptr;
        stof_io::unmix_pair(
            Z, xr, x2, n, tid, T,
            [&](int q, const float (&xa)[4], const float (&v1)[4], const float (&xb)[4], const float (&v2)[4]) {
                using stof_io::envelope;
                if (e1) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f ea = {envelope(xa[0], v1[0]), envelope(xa[1], v1[1]), envelope(xa[2], v1[2]), envelope(xa[3], v1[3])};
                    const v4f eb = {envelope(xb[0], v2[0]), envelope(xb[1], v2[1]), envelope(xb[2], v2[2]), envelope(xb[3], v2[3])};
                    if constexpr (NT) {
                        __builtin_nontemporal_store(ea, reinterpret_cast<v4f*>(e1 + 4 * q));
                        if (second) __builtin_nontemporal_store(eb, reinterpret_cast<v4f*>(e1 + n + 4 * q));
                    } else {
                        *reinterpret_cast<v4f*>(e1 + 4 * q) = ea;
                        if (second) *reinterpret_cast<v4f*>(e1 + n + 4 * q) = eb;
                    }
                }
                if (r1) {
                    *reinterpret_cast<float4*>(r1 + 4 * q) = make_float4(xa[0], xa[1], xa[2], xa[3]);
                    if (second) *reinterpret_cast<float4*>(r1 + n + 4 * q) = make_float4(xb[0], xb[1], xb[2], xb[3]);
                }
                if (i1) {
                    *reinterpret_cast<float4*>(i1 + 4 * q) = make_float4(v1[0], v1[1], v1[2], v1[3]);
                    if (second) *reinterpret_cast<float4*>(i1 + n + 4 * q) = make_float4(v2[0], v2[1], v2[2], v2[3]);
                }
            },
            [&](int i, float xa, float v1, float xb, float v2) {
                if (e1) { e1[i] = stof_io::envelope(xa, v1); if (second) e1[n + i] = stof_io::envelope(xb, v2); }
                if (r1) { r1[i] = xa; if (second) r1[n + i] = xb; }
                if (i1) { i1[i] = v1; if (second) i1[n + i] = v2; }
            });
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Compile-time plans for the row lengths the reference's datasets produce (fft_small.h: analytic_ct).  A work-group
// holds PPW pairs (one padded slot each, WPP waves per slot) and ONE twiddle table (the first N / min-radix powers of
// w_N) copied from a compile-time constant array; with WPP = 1 a wave owns its slot and no barrier separates the
// passes, only wave_lds_sync() (LDS accesses of one wave execute in order; the compiler must be told).  The pair's rows stay in registers from the load to the
// un-mixing (read from HBM once); the next pair's rows are requested before the stores of the current one drain.
// Measured on [4096, 2000] (rocprofv3, kernel only): 24.2 us against 41.8 us for the run-time plan kernel below;
// phases: launch + table 3.0, rows in 4.4, transform 13.3, un-mixing + envelope + stores 3.5 us.
// ----------------------------------------------------------------------------------------------------------------
using stof_ct::CtTwiddles;

// PAD / TW2: fft_small.h's CtOpt (rows whose padded image or full twiddle table would not fit LDS go without)
// KEEP: the pair's rows stay in registers from the load to the un-mixing (otherwise they are read a second time)
// NT: the envelope is written with non-temporal stores (a compile-time choice: behind a run-time flag the compiler merges
// the two stores into a plain one)
template <int N, int WPP, int PPW, bool PAD = true, bool TW2 = false, bool KEEP = true, bool NT = false>
__global__ __launch_bounds__(64 * WPP * PPW) void hilbert_ct_kernel(const float* __restrict__ x, long long nrows,
                                                                     float* __restrict__ env, float* __restrict__ re,
                                                                     float* __restrict__ im) {
    using namespace stof_fft;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    constexpr int T = 64 * WPP, TT = T * PPW, IO = (N / 4 + T - 1) / T;
    constexpr int TWP = TW2 ? stof_ct::twiddle2_lds_entries<N>() : stof_ct::twiddle_lds_entries<N>(), SLOT = PAD ? ct_slot_entries(N) : N;
    cf* const W = reinterpret_cast<cf*>(lds);
    const int slot = threadIdx.x / T, tid = threadIdx.x % T;
    cf* const Z = W + TWP + slot * SLOT;
    const long long npairs = (nrows + 1) / 2, stride = (long long)gridDim.x * PPW;
    long long pr = (long long)blockIdx.x * PPW + slot;

    // rows of the first pair on their way while the twiddle table is copied
    static_assert(KEEP || !PAD, "the direct HBM -> LDS staging writes the unpadded image");
    stof_io::PairRegs<KEEP ? IO : 1> cur;
    auto fetch = [&](stof_io::PairRegs<KEEP ? IO : 1>& r, long long p) {
        if (KEEP && p < npairs) {
            const float* xr = x + 2 * p * (size_t)N;
            stof_io::load_pair_regs(r, xr, 2 * p + 1 < nrows ? xr + N : nullptr, N, tid, T);
        }
    };
    fetch(cur, pr);
    if constexpr (TW2) stof_ct::stage_twiddles2<N>(lds, threadIdx.x, TT);
    else stof_ct::stage_twiddles<N>(lds, threadIdx.x, TT);
    __syncthreads();
    for (long long p0 = (long long)blockIdx.x * PPW; p0 < npairs; p0 += stride, pr += stride) {
        const bool active = pr < npairs;
        if (WPP == 1 && !active) break;                          // a lone wave: nobody waits for it
        auto sync = [] { if (WPP > 1) __syncthreads(); else wave_lds_sync(); };
        sync();                                                   // previous pair fully read back
        const float* const xrow = x + 2 * pr * (size_t)N;
        const float* const xrow2 = 2 * pr + 1 < nrows ? xrow + N : nullptr;
        if (active) {
            if constexpr (KEEP) stof_io::stage_pair<IO, PAD>(Z, cur, N, tid, T);
            else stof_io::load_pair<4>(Z, xrow, xrow2, N, tid, T);
        }
        sync();
        analytic_ct<N, T, CtOpt<PAD, TW2>>(Z, W, tid, sync);      // idle slots of a multi-wave group keep the barrier count
        if (!active) continue;
        const long long row = 2 * pr;
        const bool second = row + 1 < nrows;
        float* const e1 = env ? env + row * (size_t)N : nullptr;
        float* const r1 = re ? re + row * (size_t)N : nullptr;
        float* const i1 = im ? im + row * (size_t)N : nullptr;
        auto emit4 = [&](int q, const float (&xa)[4], const float (&v1)[4], const float (&xb)[4], const float (&v2)[4]) {
            using stof_io::envelope;
            if (e1) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f ea = {envelope(xa[0], v1[0]), envelope(xa[1], v1[1]), envelope(xa[2], v1[2]), envelope(xa[3], v1[3])};
                const v4f eb = {envelope(xb[0], v2[0]), envelope(xb[1], v2[1]), envelope(xb[2], v2[2]), envelope(xb[3], v2[3])};
                if constexpr (NT) {                            // write-once output nobody reads soon: keep it out of the caches
                    __builtin_nontemporal_store(ea, reinterpret_cast<v4f*>(e1 + 4 * q));
                    if (second) __builtin_nontemporal_store(eb, reinterpret_cast<v4f*>(e1 + N + 4 * q));
                } else {
                    *reinterpret_cast<v4f*>(e1 + 4 * q) = ea;
                    if (second) *reinterpret_cast<v4f*>(e1 + N + 4 * q) = eb;
                }
            }
            if (r1) {
                *reinterpret_cast<float4*>(r1 + 4 * q) = make_float4(xa[0], xa[1], xa[2], xa[3]);
                if (second) *reinterpret_cast<float4*>(r1 + N + 4 * q) = make_float4(xb[0], xb[1], xb[2], xb[3]);
            }
            if (i1) {
                *reinterpret_cast<float4*>(i1 + 4 * q) = make_float4(v1[0], v1[1], v1[2], v1[3]);
                if (second) *reinterpret_cast<float4*>(i1 + N + 4 * q) = make_float4(v2[0], v2[1], v2[2], v2[3]);
            }
        };
        if constexpr (KEEP) stof_io::unmix_pair_regs<IO, PAD>(Z, cur, N, tid, T, emit4);
        else stof_io::unmix_pair<4>(Z, xrow, xrow2, N, tid, T, emit4, [](int, float, float, float, float) {});   // (rows are 16-byte aligned here)
        fetch(cur, pr + stride);                                   // next pair's rows (its latency hides behind the stores)
    }
}

template <int N, int WPP, int PPW, bool PAD = true, bool TW2 = false, bool KEEP = true>
int launch_ct(const float* x, int64_t nrows, float* env, float* re, float* im, int ncu, hipStream_t stream, int streamed) {
    constexpr size_t lds = ((size_t)(TW2 ? stof_ct::twiddle2_lds_entries<N>() : stof_ct::twiddle_lds_entries<N>()) +
                            (size_t)PPW * (PAD ? stof_fft::ct_slot_entries(N) : N)) * sizeof(float2);
    static_assert(lds <= (size_t)LDS_BYTES, "slots + table exceed LDS");
    static stof::LdsLimitOnce once[2];
    const void* const kern = streamed ? reinterpret_cast<const void*>(&hilbert_ct_kernel<N, WPP, PPW, PAD, TW2, KEEP, true>)
                                      : reinterpret_cast<const void*>(&hilbert_ct_kernel<N, WPP, PPW, PAD, TW2, KEEP, false>);
    if (int st = once[streamed ? 1 : 0].ensure(kern, LDS_BYTES)) return st;
    const int64_t npairs = (nrows + 1) / 2, groups = (npairs + PPW - 1) / PPW;
    int64_t grid = (int64_t)ncu * (int64_t)((size_t)LDS_BYTES / lds);
    if (grid > groups) grid = groups;
    if (streamed)
        hipLaunchKernelGGL((hilbert_ct_kernel<N, WPP, PPW, PAD, TW2, KEEP, true>), dim3((unsigned)grid), dim3(64 * WPP * PPW), lds,
                           stream, x, (long long)nrows, env, re, im);
    else
        hipLaunchKernelGGL((hilbert_ct_kernel<N, WPP, PPW, PAD, TW2, KEEP, false>), dim3((unsigned)grid), dim3(64 * WPP * PPW), lds,
                           stream, x, (long long)nrows, env, re, im);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// lengths with a compile-time plan; returns -1 if n has none (or STOF_HILBERT_CT=0 asks for the run-time plan)
int try_launch_ct(const float* x, int64_t nrows, int64_t n, float* env, float* re, float* im, int ncu, hipStream_t stream, int streamed) {
    static const int mode = [] { const char* e = getenv("STOF_HILBERT_CT"); return e ? atoi(e) : 1; }();
    if (mode == 0) return -1;
    for (const void* p : {(const void*)x, (const void*)env, (const void*)re, (const void*)im})
        if (reinterpret_cast<size_t>(p) & 15) return -1;        // 16-byte row accesses
    switch (n) {
        // one wave per pair up to 2048 samples (16 register-resident 16-byte pieces per lane), two waves beyond
        case 1536: return launch_ct<1536, 1, 4>(x, nrows, env, re, im, ncu, stream, streamed);
        case 2000: return launch_ct<2000, 1, 4>(x, nrows, env, re, im, ncu, stream, streamed);
        case 2048: return launch_ct<2048, 1, 4>(x, nrows, env, re, im, ncu, stream, streamed);
        case 4000: return launch_ct<4000, 2, 2>(x, nrows, env, re, im, ncu, stream, streamed);
        case 4096: return launch_ct<4096, 2, 2>(x, nrows, env, re, im, ncu, stream, streamed);
        case 6144: return launch_ct<6144, 4, 2>(x, nrows, env, re, im, ncu, stream, streamed);          // PALA frames (1536) x rf 4
        case 8000: return launch_ct<8000, 4, 2>(x, nrows, env, re, im, ncu, stream, streamed);
        case 15360: return launch_ct<15360, 8, 1>(x, nrows, env, re, im, ncu, stream, streamed);        // PALA frames x rf 10
        // 20,000 values = 160,000 bytes: the image alone nearly fills LDS -> unpadded, two-level twiddles, eight waves per pair
        case 20000: return launch_ct<20000, 8, 1, false, true, false>(x, nrows, env, re, im, ncu, stream, streamed);
        default: return -1;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Rows beyond LDS whose length factors as n = R0 * M with M a compile-time-plan length: four-step transform.
//   outer_forward_kernel<R0>    thread j: z_k = (x1, x2)[j + M k], y = DFT_R0(z), y_q *= w_n^{j q}  -> scratch[pair][q][j]
//   hilbert_ct_blocks_kernel<M> every block (pair, q) is one length-M analytic_ct in LDS: DFT_M(y_q)[k'] is the frequency
//                               q + R0 k', so the block filters with the full length's scales and only block 0 holds the
//                               DC and Nyquist bins (CtFilter)
//   outer_inverse_kernel<R0>    thread j: y_q conj(w_n^{j q}), inverse DFT_R0 -> analytic pair at j + M k; un-mixing with
//                               the rows and envelope / re / im stores
// The scratch (8 n bytes per pair) lives in the workspace; rows are processed in chunks of <= FOURSTEP_CHUNK_BYTES of
// scratch so that it stays in the 256 MB Infinity Cache between the three kernels.
// ----------------------------------------------------------------------------------------------------------------
constexpr size_t FOURSTEP_CHUNK_BYTES = 64u << 20;

__global__ void outer_table_kernel(float2* __restrict__ tw, int M, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    double s, c;
    sincospi(-2.0 * (double)j / (double)n, &s, &c);
    tw[j] = make_float2((float)c, (float)s);
}

// w[q] = w1^q, q = 1 .. R-1 (w^2q = (w^q)^2, w^(2q+1) = w^2q w)
template <int R>
__device__ __forceinline__ void twiddle_powers(stof_fft::cf w1, stof_fft::cf (&w)[R]) {
    using namespace stof_fft;
    w[1] = w1;
    static_for<R>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q >= 2) w[q] = (q % 2 == 0) ? stof_fft::cmul(w[q / 2], w[q / 2]) : stof_fft::cmul(w[q - 1], w[1]);
    });
}

template <int R0>
__global__ __launch_bounds__(256) void outer_forward_kernel(const float* __restrict__ x, long long nrows, long long pair0, int M,
                                                            const float2* __restrict__ twn, float2* __restrict__ scratch) {
    using namespace stof_fft;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const long long pair = blockIdx.y, row = 2 * (pair0 + pair);
    const size_t n = (size_t)R0 * M;
    const float* x1 = x + row * n;
    const float* x2 = row + 1 < nrows ? x1 + n : nullptr;
    cf z[R0];
#pragma unroll
    for (int k = 0; k < R0; ++k) z[k] = mk(x1[j + (size_t)M * k], x2 ? x2[j + (size_t)M * k] : 0.f);
    Bf<R0>::run(z);
    cf w[R0];
    const float2 t = twn[j];
    twiddle_powers<R0>(mk(t.x, t.y), w);
    float2* out = scratch + (size_t)pair * n + j;
    out[0] = make_float2(z[0].x, z[0].y);
#pragma unroll
    for (int q = 1; q < R0; ++q) {
        const cf v = stof_fft::cmul(z[q], w[q]);
        out[(size_t)q * M] = make_float2(v.x, v.y);
    }
}

template <int R0, bool NT>
__global__ __launch_bounds__(256) void outer_inverse_kernel(const float* __restrict__ x, long long nrows, long long pair0, int M,
                                                            const float2* __restrict__ twn, const float2* __restrict__ scratch,
                                                            float* __restrict__ env, float* __restrict__ re, float* __restrict__ im) {
    using namespace stof_fft;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const long long pair = blockIdx.y, row = 2 * (pair0 + pair);
    const size_t n = (size_t)R0 * M;
    const bool second = row + 1 < nrows;
    const float* x1 = x + row * n;
    const float* x2 = second ? x1 + n : nullptr;
    const float2* in = scratch + (size_t)pair * n + j;
    cf y[R0], w[R0];
    const float2 t = twn[j];
    twiddle_powers<R0>(mk(t.x, t.y), w);
#pragma unroll
    for (int q = 0; q < R0; ++q) {
        const float2 v = in[(size_t)q * M];
        y[q] = q ? stof_fft::cmulc(mk(v.x, v.y), w[q]) : mk(v.x, v.y);
    }
    Bf<R0>::run(y);                                               // inverse DFT = forward, outputs reversed
#pragma unroll
    for (int k = 0; k < R0; ++k) {
        const cf z = y[(R0 - k) % R0];
        const size_t i = j + (size_t)M * k;
        const float a = x1[i], b = x2 ? x2[i] : 0.f;
        const float v1 = z.y - b, v2 = a - z.x;                   // pair un-mixing (see hilbert_pairs_kernel)
        if (env) {
            const float ea = stof_io::envelope(a, v1), eb = stof_io::envelope(b, v2);
            if constexpr (NT) {
                __builtin_nontemporal_store(ea, env + row * n + i);
                if (second) __builtin_nontemporal_store(eb, env + (row + 1) * n + i);
            } else {
                env[row * n + i] = ea;
                if (second) env[(row + 1) * n + i] = eb;
            }
        }
        if (re) { re[row * n + i] = a; if (second) re[(row + 1) * n + i] = b; }
        if (im) { im[row * n + i] = v1; if (second) im[(row + 1) * n + i] = v2; }
    }
}

template <int M, int WPP, int PPW>
__global__ __launch_bounds__(64 * WPP * PPW) void hilbert_ct_blocks_kernel(float2* __restrict__ blocks, long long nblocks, int R0,
                                                                            float inv_n) {
    using namespace stof_fft;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    constexpr int T = 64 * WPP, TT = T * PPW;
    constexpr int TWP = stof_ct::twiddle_lds_entries<M>(), SLOT = ct_slot_entries(M);
    cf* const W = reinterpret_cast<cf*>(lds);
    const int slot = threadIdx.x / T, tid = threadIdx.x % T;
    cf* const Z = W + TWP + slot * SLOT;
    stof_ct::stage_twiddles<M>(lds, threadIdx.x, TT);
    __syncthreads();
    const long long stride = (long long)gridDim.x * PPW;
    auto sync = [] { if (WPP > 1) __syncthreads(); else wave_lds_sync(); };
    for (long long b0 = (long long)blockIdx.x * PPW; b0 < nblocks; b0 += stride) {
        const long long b = b0 + slot;
        const bool active = b < nblocks;
        if (WPP == 1 && !active) break;
        stof_io::f4a* const blk = reinterpret_cast<stof_io::f4a*>(blocks + (active ? b : 0) * (long long)M);
        sync();
        if (active) {
            for (int i = tid; i < M / 2; i += T)                  // two complex values per 16-byte piece; 2 i is even, so the
                *reinterpret_cast<stof_io::f4a*>(Z + 2 * i + 2 * (i >> 3)) = blk[i];      // pair stays inside its 16-block
        }
        sync();
        CtFilter filt;
        filt.one = inv_n; filt.two = 2.0f * inv_n; filt.edge = (b % R0) == 0;
        analytic_ct<M, T>(Z, W, tid, sync, filt);
        if (active) {
            for (int i = tid; i < M / 2; i += T) blk[i] = *reinterpret_cast<const stof_io::f4a*>(Z + 2 * i + 2 * (i >> 3));
        }
    }
}

struct FourStepPlan { int R0, M; };
bool four_step_plan(int64_t n, FourStepPlan* p) {
    static const int ms[5] = {4096, 4000, 2048, 2000, 1536};
    // (n = R0 * M only matters beyond the LDS fast path, n > 20480 >= 5 * 4096, so the outer radix is at least 6; for rows that
    // fit LDS the single in-LDS kernel wins: [1024,20000] 119 us against 132 us as 10 x 2000)
    static const int rs[6] = {6, 8, 10, 15, 16, 20};
    for (int m : ms) {
        if (n % m) continue;
        const int64_t r = n / m;
        for (int c : rs) if (r == c) { p->R0 = c; p->M = m; return true; }
    }
    return false;
}
int64_t four_step_chunk_pairs(int64_t n) {
    const int64_t c = (int64_t)(FOURSTEP_CHUNK_BYTES / ((size_t)n * sizeof(float2)));
    return c < 1 ? 1 : c;
}

template <int M, int WPP, int PPW>
int launch_blocks(float2* blocks, int64_t nblocks, int R0, float inv_n, int ncu, hipStream_t stream) {
    constexpr size_t lds = ((size_t)stof_ct::twiddle_lds_entries<M>() + (size_t)PPW * stof_fft::ct_slot_entries(M)) * sizeof(float2);
    static stof::LdsLimitOnce once;
    if (int st = once.ensure(reinterpret_cast<const void*>(&hilbert_ct_blocks_kernel<M, WPP, PPW>), LDS_BYTES)) return st;
    const int64_t groups = (nblocks + PPW - 1) / PPW;
    int64_t grid = (int64_t)ncu * (int64_t)((size_t)LDS_BYTES / lds);
    if (grid > groups) grid = groups;
    hipLaunchKernelGGL((hilbert_ct_blocks_kernel<M, WPP, PPW>), dim3((unsigned)grid), dim3(64 * WPP * PPW), lds, stream, blocks,
                       (long long)nblocks, R0, inv_n);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

template <int R0>
void launch_outer(bool inverse, const float* x, int64_t nrows, int64_t pair0, int64_t npairs, int M, const float2* twn,
                  float2* scratch, float* env, float* re, float* im, hipStream_t stream, int streamed) {
    const dim3 grid((unsigned)((M + 255) / 256), (unsigned)npairs);
    if (inverse && streamed)
        hipLaunchKernelGGL((outer_inverse_kernel<R0, true>), grid, dim3(256), 0, stream, x, (long long)nrows, (long long)pair0, M, twn,
                           (const float2*)scratch, env, re, im);
    else if (inverse)
        hipLaunchKernelGGL((outer_inverse_kernel<R0, false>), grid, dim3(256), 0, stream, x, (long long)nrows, (long long)pair0, M, twn,
                           (const float2*)scratch, env, re, im);
    else
        hipLaunchKernelGGL(outer_forward_kernel<R0>, grid, dim3(256), 0, stream, x, (long long)nrows, (long long)pair0, M, twn, scratch);
}

// the workspace holds w_n^j (j < M) followed by the scratch of one chunk of pairs
int run_four_step(const FourStepPlan& fp, const float* x, int64_t nrows, int64_t n, float* env, float* re, float* im,
                  void* workspace, int ncu, hipStream_t stream, int streamed) {
    float2* twn = static_cast<float2*>(workspace);
    float2* scratch = reinterpret_cast<float2*>(reinterpret_cast<char*>(workspace) + ((size_t)fp.M * sizeof(float2) + 255) / 256 * 256);
    hipLaunchKernelGGL(outer_table_kernel, dim3((unsigned)((fp.M + 255) / 256)), dim3(256), 0, stream, twn, fp.M, (int)n);
    const int64_t npairs = (nrows + 1) / 2, chunk = four_step_chunk_pairs(n);
    const float inv_n = (float)(1.0 / (double)n);
    for (int64_t p0 = 0; p0 < npairs; p0 += chunk) {
        const int64_t np = npairs - p0 < chunk ? npairs - p0 : chunk;
        for (int pass = 0; pass < 3; ++pass) {
            if (pass == 1) {
                int st;
                switch (fp.M) {
                    case 1536: st = launch_blocks<1536, 1, 4>(scratch, np * fp.R0, fp.R0, inv_n, ncu, stream); break;
                    case 2000: st = launch_blocks<2000, 1, 4>(scratch, np * fp.R0, fp.R0, inv_n, ncu, stream); break;
                    case 2048: st = launch_blocks<2048, 1, 4>(scratch, np * fp.R0, fp.R0, inv_n, ncu, stream); break;
                    case 4000: st = launch_blocks<4000, 2, 2>(scratch, np * fp.R0, fp.R0, inv_n, ncu, stream); break;
                    default: st = launch_blocks<4096, 2, 2>(scratch, np * fp.R0, fp.R0, inv_n, ncu, stream); break;
                }
                if (st != STOF_OK) return st;
                continue;
            }
            const bool inv = pass == 2;
            switch (fp.R0) {
                case 6: launch_outer<6>(inv, x, nrows, p0, np, fp.M, twn, scratch, env, re, im, stream, streamed); break;
                case 8: launch_outer<8>(inv, x, nrows, p0, np, fp.M, twn, scratch, env, re, im, stream, streamed); break;
                case 10: launch_outer<10>(inv, x, nrows, p0, np, fp.M, twn, scratch, env, re, im, stream, streamed); break;
                case 15: launch_outer<15>(inv, x, nrows, p0, np, fp.M, twn, scratch, env, re, im, stream, streamed); break;
                case 16: launch_outer<16>(inv, x, nrows, p0, np, fp.M, twn, scratch, env, re, im, stream, streamed); break;
                default: launch_outer<20>(inv, x, nrows, p0, np, fp.M, twn, scratch, env, re, im, stream, streamed); break;
            }
        }
    }
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

}  // namespace

namespace stof {
// LDS bytes of the fast path for rows of n samples, 0 if n has no plan or does not fit (shared with gradpeak.hip)
size_t hilbert_fast_lds_bytes(int64_t n, stof_fft::Plan* plan_out) {
    if (n < 2 || n > LDS_BYTES / 8) return 0;
    stof_fft::Plan p;
    if (!stof_fft::make_plan((int)n, &p)) return 0;
    const size_t bytes = ((size_t)n + stof_fft::twiddle_entries((int)n)) * sizeof(float2);
    if (bytes > (size_t)LDS_BYTES) return 0;
    if (plan_out) *plan_out = p;
    return bytes;
}
}  // namespace stof

extern "C" size_t stof_hilbert_workspace_bytes(int64_t N, int64_t n) {
    if (n <= 0 || N <= 0) return 0;
    if (stof::hilbert_fast_lds_bytes(n, nullptr)) return 256;                     // fast path: tables live in LDS
    size_t bytes = (size_t)n * (sizeof(float2) + sizeof(float)) + 256;             // twiddle + filter tables
    if ((size_t)n * sizeof(float2) > (size_t)LDS_BYTES / 2 || n > LDS_BYTES / 8) {
        // rows that may not fit LDS (two buffers when a prime radix > 5 is present): per-work-group global scratch
        const int64_t units = (n % 2 == 0) ? (N + 1) / 2 : N;
        const int64_t grid = units < GENERIC_ZG_GRID ? units : GENERIC_ZG_GRID;
        bytes += (size_t)grid * 2 * (size_t)n * sizeof(float2);
    }
    FourStepPlan fsp;
    if (n % 2 == 0 && four_step_plan(n, &fsp)) {                   // n = R0 * M: twiddle table + the scratch of one chunk of pairs
        const int64_t npairs = (N + 1) / 2, chunk = four_step_chunk_pairs(n);
        const size_t fs = ((size_t)fsp.M * sizeof(float2) + 255) / 256 * 256 +
                          (size_t)(npairs < chunk ? npairs : chunk) * (size_t)n * sizeof(float2) + 256;
        if (fs > bytes) bytes = fs;                                // (the generic kernels remain the fallback for unaligned rows)
    }
    return bytes;
}

namespace {
int hilbert_impl(const float* x, int64_t N, int64_t n, float* env, float* re, float* im, void* workspace, size_t workspace_bytes,
                 void* stream_, int streamed);
}

extern "C" int stof_hilbert(const float* x, int64_t N, int64_t n, float* env, float* re, float* im,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return hilbert_impl(x, N, n, env, re, im, workspace, workspace_bytes, stream, 0);
}

extern "C" int stof_hilbert_streamed(const float* x, int64_t N, int64_t n, float* env, float* re, float* im,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return hilbert_impl(x, N, n, env, re, im, workspace, workspace_bytes, stream, 1);
}

namespace {
int hilbert_impl(const float* x, int64_t N, int64_t n, float* env, float* re, float* im, void* workspace, size_t workspace_bytes,
                 void* stream_, int streamed) {
    if (N < 0 || n < 0) return STOF_ERR_BAD_ARG;
    if (N == 0 || n == 0) return STOF_OK;
    if (!x || (!env && !re && !im)) return STOF_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < stof_hilbert_workspace_bytes(N, n)) return STOF_ERR_WORKSPACE;
    if (N > 0x7fffffffLL || n > (1 << 22)) return STOF_ERR_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int ncu = stof::device_cu_count();

    {
        const int st = try_launch_ct(x, N, n, env, re, im, ncu, stream, streamed);
        if (st >= 0) return st;
    }
    stof_fft::Plan fplan;
    if (const size_t flds = stof::hilbert_fast_lds_bytes(n, &fplan)) {
        static stof::LdsLimitOnce fast_once;
        if (int st = fast_once.ensure(reinterpret_cast<const void*>(&hilbert_pairs_kernel<false>), LDS_BYTES)) return st;
        static stof::LdsLimitOnce fast_once_nt;
        if (int st = fast_once_nt.ensure(reinterpret_cast<const void*>(&hilbert_pairs_kernel<true>), LDS_BYTES)) return st;
        // threads per pair: enough waves per CU to hide LDS latency whatever the row length
        int per_cu = (int)((size_t)LDS_BYTES / flds);
        if (per_cu > 16) per_cu = 16;
        int threads = 64;                                          // 16 waves per CU whatever the row length
        while (threads < 512 && per_cu * (threads / 64) < 16) threads *= 2;
        if (const char* e = getenv("STOF_HILBERT_THREADS")) threads = atoi(e);       // tuning / A-B switch
        const int64_t npairs = (N + 1) / 2;
        int64_t grid = (int64_t)ncu * per_cu;
        if (grid > npairs) grid = npairs;
        if (streamed)
            hipLaunchKernelGGL(hilbert_pairs_kernel<true>, dim3((unsigned)grid), dim3(threads), flds, stream, x, fplan, (long long)N,
                           env, re, im);
        else
            hipLaunchKernelGGL(hilbert_pairs_kernel<false>, dim3((unsigned)grid), dim3(threads), flds, stream, x, fplan, (long long)N,
                           env, re, im);
        return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
    }

    {
        static const int fs_mode = [] { const char* e = getenv("STOF_HILBERT_FOURSTEP"); return e ? atoi(e) : 1; }();
        FourStepPlan fsp;
        bool aligned = true;
        for (const void* p : {(const void*)x, (const void*)env, (const void*)re, (const void*)im, (const void*)workspace})
            aligned = aligned && !(reinterpret_cast<size_t>(p) & 15);
        if (fs_mode && aligned && n % 2 == 0 && four_step_plan(n, &fsp))
            return run_four_step(fsp, x, N, n, env, re, im, workspace, ncu, stream, streamed);
    }
    FftPlan plan;
    make_plan((int)n, &plan);
    const size_t data_lds = (size_t)n * sizeof(float2) * (plan.needs_second ? 2 : 1);
    const bool zg = data_lds > (size_t)LDS_BYTES;                                    // row (pair) beyond LDS: global scratch
    const bool tw_lds = !zg && data_lds + (size_t)n * sizeof(float2) <= (size_t)LDS_BYTES;
    const size_t lds = zg ? 0 : data_lds + (tw_lds ? (size_t)n * sizeof(float2) : 0);
    const bool pair = (n % 2 == 0);
    using Kern = void (*)(const float*, const float2*, const float*, const FftPlan, long long, float*, float*, float*, float2*);
    const Kern kern = zg ? (pair ? &hilbert_generic_kernel<false, true, true> : &hilbert_generic_kernel<false, false, true>)
                    : tw_lds ? (pair ? &hilbert_generic_kernel<true, true, false> : &hilbert_generic_kernel<true, false, false>)
                             : (pair ? &hilbert_generic_kernel<false, true, false> : &hilbert_generic_kernel<false, false, false>);
    if (!zg) {
        static stof::LdsLimitOnce lds_once[4];
        if (int st = lds_once[(tw_lds ? 2 : 0) + (pair ? 1 : 0)].ensure(reinterpret_cast<const void*>(kern), LDS_BYTES)) return st;
    }
    float2* tw = static_cast<float2*>(workspace);
    float* hf = reinterpret_cast<float*>(tw + n);
    float2* scratch = reinterpret_cast<float2*>(reinterpret_cast<char*>(workspace) +
                                                ((size_t)n * (sizeof(float2) + sizeof(float)) + 255) / 256 * 256);
    hipLaunchKernelGGL(tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tw, hf, plan);
    const int64_t units = pair ? (N + 1) / 2 : N;
    int64_t grid;
    if (zg) {
        grid = units < GENERIC_ZG_GRID ? units : GENERIC_ZG_GRID;
    } else {
        // persistent-ish grid: as many work-groups as can be resident (LDS-limited), each looping over rows
        int64_t per_cu = (int64_t)LDS_BYTES / (int64_t)(lds ? lds : 1);
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 4) per_cu = 4;                                      // 512 threads x 4 = 32 waves per CU
        grid = ncu * per_cu;
        if (grid > units) grid = units;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, stream, x, tw, hf, plan, (long long)N, env, re, im, scratch);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
}  // namespace

// ----------------------------------------------------------------------------------------------------------------
// float64 rows (utils/hilbert.py:5-21 keeps a float64 input in complex128: torch.fft.fft follows the input's dtype).
// Not a tuned path -- no configuration of the reference feeds float64 -- but the same O(n log n) transform for any n:
// one work-group per row walks a mixed-radix Stockham autosort plan (decimation in frequency; radices = the prime
// factors of n, 4 where two 2s pair up) between two global scratch rows, twiddles from a table e^{-2 pi i j / n}
// built in double with sincospi (exact argument reduction: j / n is formed once, no accumulated angles).
//   stage (radix R, current length c = R m, stride s, c s = n):  out[q + s (R p + k)] = W_c^{p k} sum_r in[q + s (p + r m)] W_R^{r k}
//   (p < m, k < R, q < s); W_c^{p k} = tw[p k s] (p k s < n), W_R^{r k} = tw[((r k) mod R) n / R].
// The inverse runs the same plan on the conjugated, filtered spectrum (ifft(F) = conj(fft(conj(F))) / n).
// ----------------------------------------------------------------------------------------------------------------
namespace {
constexpr int F64_MAX_FACTORS = 32, F64_GRID = 512, F64_THREADS = 256;
struct F64Plan { int nf; int radix[F64_MAX_FACTORS]; };

__global__ void hilbert_f64_twiddle_kernel(double2* tw, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double sn, cs;
    sincospi(2.0 * (double)j / (double)n, &sn, &cs);
    tw[j] = make_double2(cs, -sn);
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// one forward transform of the row held in `a` (natural order); returns the buffer that holds the result
__device__ double2* f64_forward(double2* a, double2* b, const double2* __restrict__ tw, const int n, const F64Plan& plan) {
    int c = n, s = 1;
    for (int f = 0; f < plan.nf; ++f) {
        const int R = plan.radix[f], m = c / R, step = n / R;
        for (int o = threadIdx.x; o < n; o += blockDim.x) {
            const int q = o % s, t = o / s, k = t % R, p = t / R;
            const double2* src = a + q + (long long)s * p;
            double2 acc = src[0];
            int rk = 0;                                                     // (r k) mod R
            for (int r = 1; r < R; ++r) {
                rk += k;
                if (rk >= R) rk -= R;
                const double2 v = src[(long long)s * m * r], w = tw[(long long)rk * step];
                acc.x += v.x * w.x - v.y * w.y;
                acc.y += v.x * w.y + v.y * w.x;
            }
            b[o] = cmul(acc, tw[(long long)p * k * s]);
        }
        __syncthreads();
        double2* const tmp = a; a = b; b = tmp;
        c = m; s *= R;
    }
    return a;
}

__global__ __launch_bounds__(F64_THREADS) void hilbert_f64_kernel(const double* __restrict__ x, const long long N, const int n,
                                                                    double* env, double* re, double* im,
                                                                    const double2* __restrict__ tw, double2* scratch, const F64Plan plan,
                                                                    const int use_lds) {
    // rows of up to LDS_BYTES / 32 samples (5,120) keep both buffers in LDS (flat addressing: the same code walks either)
    extern __shared__ __attribute__((aligned(16))) unsigned char f64_rows[];
    double2* const bufA = use_lds ? reinterpret_cast<double2*>(f64_rows) : scratch + (long long)blockIdx.x * 2 * n;
    double2* const bufB = bufA + n;
    const int half = n / 2;
    const double inv_n = 1.0 / (double)n;
    for (long long row = blockIdx.x; row < N; row += gridDim.x) {
        const double* xr = x + row * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) bufA[i] = make_double2(xr[i], 0.0);
        __syncthreads();
        double2* f = f64_forward(bufA, bufB, tw, n, plan);
        double2* other = f == bufA ? bufB : bufA;
        // utils/hilbert.py:13-17: bins above n // 2 zeroed, bins 1 .. n // 2 - 1 doubled (bin n // 2 kept as it is, also for
        // odd n: Q6); conjugated for the inverse
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const double h = k > half ? 0.0 : ((k >= 1 && k < half) ? 2.0 : 1.0);
            const double2 v = f[k];
            f[k] = make_double2(h * v.x, -h * v.y);
        }
        __syncthreads();
        const double2* v = f64_forward(f, other, tw, n, plan);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double a = v[i].x * inv_n, b = -v[i].y * inv_n;
            if (re) re[row * n + i] = a;
            if (im) im[row * n + i] = b;
            if (env) env[row * n + i] = hypot(a, b);
        }
        __syncthreads();
    }
}

bool f64_plan(int64_t n, F64Plan* plan) {
    plan->nf = 0;
    int twos = 0;
    while (n % 2 == 0) { ++twos; n /= 2; }
    for (; twos >= 2; twos -= 2) plan->radix[plan->nf++] = 4;
    if (twos) plan->radix[plan->nf++] = 2;
    for (int64_t d = 3; d * d <= n; d += 2)
        while (n % d == 0) {
            if (plan->nf >= F64_MAX_FACTORS) return false;
            plan->radix[plan->nf++] = (int)d;
            n /= d;
        }
    if (n > 1) {
        if (plan->nf >= F64_MAX_FACTORS) return false;
        plan->radix[plan->nf++] = (int)n;
    }
    return true;
}
}  // namespace

extern "C" size_t stof_hilbert_f64_workspace_bytes(int64_t N, int64_t n) {
    if (n <= 0 || N <= 0) return 0;
    const int64_t grid = N < F64_GRID ? N : F64_GRID;
    return (size_t)n * sizeof(double2) * (1 + 2 * (size_t)grid) + 256;
}

extern "C" int stof_hilbert_f64(const double* x, int64_t N, int64_t n, double* env, double* re, double* im,
                                void* workspace, size_t workspace_bytes, void* stream_) {
    if (N < 0 || n < 0) return STOF_ERR_BAD_ARG;
    if (N == 0 || n == 0) return STOF_OK;
    if (!x || (!env && !re && !im)) return STOF_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < stof_hilbert_f64_workspace_bytes(N, n)) return STOF_ERR_WORKSPACE;
    if (n > (1 << 22)) return STOF_ERR_UNSUPPORTED;
    F64Plan plan;
    if (!f64_plan(n, &plan)) return STOF_ERR_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double2* const tw = reinterpret_cast<double2*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    double2* const scratch = tw + n;
    const int grid = (int)(N < F64_GRID ? N : F64_GRID);
    const size_t lds = (size_t)2 * n * sizeof(double2);
    const int use_lds = lds <= (size_t)LDS_BYTES ? 1 : 0;
    if (use_lds) {
        static stof::LdsLimitOnce f64_once;
        if (int st = f64_once.ensure(reinterpret_cast<const void*>(&hilbert_f64_kernel), LDS_BYTES)) return st;
    }
    hipLaunchKernelGGL(hilbert_f64_twiddle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tw, (int)n);
    hipLaunchKernelGGL(hilbert_f64_kernel, dim3(grid), dim3(F64_THREADS), use_lds ? lds : 0, stream, x, (long long)N, (int)n, env, re, im, tw,
                       scratch, plan, use_lds);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
