// HBM <-> LDS movement of a PAIR of real rows around the complex transform of fft_small.h.
//
// The pair rides one complex transform z = x1 + i x2.  A naive `for (i) Z[i] = (x1[i], x2[i])` makes every iteration
// wait for its own HBM round trip (~1-2 us), which was most of the old kernels' run time; here a thread issues up to
// IO_UNROLL 16-byte loads per row before it touches the first result, so one latency covers the whole row.
#pragma once
#include <hip/hip_runtime.h>
#include "fft_small.h"

namespace stof_io {

using stof_fft::cf;
typedef float4 __attribute__((may_alias)) f4a;      // 16-byte access to memory that is also read / written as cf or float

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// Z[i] = (x1[i], x2[i]) for i < n; x2 = nullptr gives zeros.  IO_UNROLL: 16-byte loads in flight per row and thread.
template <int IO_UNROLL = 8>
__device__ __forceinline__ void load_pair(cf* __restrict__ Z, const float* __restrict__ x1, const float* __restrict__ x2,
                                          int n, int tid, int T) {
    if (((n & 3) == 0) && aligned16(x1) && (x2 == nullptr || aligned16(x2))) {
        const int nq = n >> 2;
        for (int q0 = tid; q0 < nq; q0 += T * IO_UNROLL) {
            float4 a[IO_UNROLL], b[IO_UNROLL];
#pragma unroll
            for (int k = 0; k < IO_UNROLL; ++k) {
                const int q = q0 + k * T;
                a[k] = b[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < nq) {
                    a[k] = *reinterpret_cast<const f4a*>(x1 + 4 * q);
                    if (x2) b[k] = *reinterpret_cast<const f4a*>(x2 + 4 * q);
                }
            }
#pragma unroll
            for (int k = 0; k < IO_UNROLL; ++k) {
                const int q = q0 + k * T;
                if (q < nq) {
                    f4a* d = reinterpret_cast<f4a*>(Z + 4 * q);
                    d[0] = make_float4(a[k].x, b[k].x, a[k].y, b[k].y);
                    d[1] = make_float4(a[k].z, b[k].z, a[k].w, b[k].w);
                }
            }
        }
    } else {
        for (int i0 = tid; i0 < n; i0 += T * IO_UNROLL) {
            float a[IO_UNROLL], b[IO_UNROLL];
#pragma unroll
            for (int k = 0; k < IO_UNROLL; ++k) {
                const int i = i0 + k * T;
                a[k] = (i < n) ? x1[i] : 0.f;
                b[k] = (i < n && x2) ? x2[i] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < IO_UNROLL; ++k) {
                const int i = i0 + k * T;
                if (i < n) Z[i] = stof_fft::mk(a[k], b[k]);
            }
        }
    }
}

// After the analytic transform Z = a1 + i a2 (a_j = x_j + i v_j): v1 = Im Z - x2, v2 = x1 - Re Z.  Calls
//   emit4(q, x1[4], v1[4], x2[4], v2[4])   for four consecutive samples 4q .. 4q+3 (vector path), or
//   emit1(i, x1, v1, x2, v2)               per sample (rows whose length or alignment rules the vector path out).
template <int IO_UNROLL = 8, class Emit4, class Emit1>
__device__ __forceinline__ void unmix_pair(const cf* __restrict__ Z, const float* __restrict__ x1, const float* __restrict__ x2,
                                           int n, int tid, int T, Emit4 emit4, Emit1 emit1) {
    if (((n & 3) == 0) && aligned16(x1) && (x2 == nullptr || aligned16(x2))) {
        const int nq = n >> 2;
        for (int q0 = tid; q0 < nq; q0 += T * IO_UNROLL) {
            float4 a[IO_UNROLL], b[IO_UNROLL];
#pragma unroll
            for (int k = 0; k < IO_UNROLL; ++k) {
                const int q = q0 + k * T;
                a[k] = b[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < nq) {
                    a[k] = *reinterpret_cast<const f4a*>(x1 + 4 * q);
                    if (x2) b[k] = *reinterpret_cast<const f4a*>(x2 + 4 * q);
                }
            }
#pragma unroll
            for (int k = 0; k < IO_UNROLL; ++k) {
                const int q = q0 + k * T;
                if (q < nq) {
                    const f4a* s = reinterpret_cast<const f4a*>(Z + 4 * q);
                    const float4 z0 = s[0], z1 = s[1];
                    const float xa[4] = {a[k].x, a[k].y, a[k].z, a[k].w}, xb[4] = {b[k].x, b[k].y, b[k].z, b[k].w};
                    const float re[4] = {z0.x, z0.z, z1.x, z1.z}, im[4] = {z0.y, z0.w, z1.y, z1.w};
                    float v1[4], v2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v1[e] = im[e] - xb[e]; v2[e] = xa[e] - re[e]; }
                    emit4(q, xa, v1, xb, v2);
                }
            }
        }
    } else {
        for (int i = tid; i < n; i += T) {
            const cf v = Z[i];
            const float a = x1[i], b = x2 ? x2[i] : 0.f;
            emit1(i, a, v.y - b, b, a - v.x);
        }
    }
}

// ---- the same movement with the rows kept in registers ---------------------------------------------------------------
// A thread's 16-byte pieces q = tid + k T (k < IO) of both rows stay in registers from the load to the un-mixing, so the
// rows are read from HBM once, and a caller can issue the NEXT pair's loads before it transforms the current one.
// Requires n % 4 == 0, 16-byte aligned rows and n <= 4 T IO.
template <int IO>
struct PairRegs {
    float4 a[IO], b[IO];
};

template <int IO>
__device__ __forceinline__ void load_pair_regs(PairRegs<IO>& r, const float* __restrict__ x1, const float* __restrict__ x2,
                                               int n, int tid, int T) {
    const int nq = n >> 2;
#pragma unroll
    for (int k = 0; k < IO; ++k) {
        const int q = tid + k * T;
        r.a[k] = r.b[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) {
            r.a[k] = *reinterpret_cast<const f4a*>(x1 + 4 * q);
            if (x2) r.b[k] = *reinterpret_cast<const f4a*>(x2 + 4 * q);
        }
    }
}

// PADDED: Z is the padded slot of fft_small.h's compile-time plans (two complex values after every 16)
template <int IO, bool PADDED = false>
__device__ __forceinline__ void stage_pair(cf* __restrict__ Z, const PairRegs<IO>& r, int n, int tid, int T) {
    const int nq = n >> 2;
#pragma unroll
    for (int k = 0; k < IO; ++k) {
        const int q = tid + k * T;
        if (q < nq) {
            f4a* d = reinterpret_cast<f4a*>(Z + (PADDED ? 4 * q + 2 * (q >> 2) : 4 * q));
            d[0] = make_float4(r.a[k].x, r.b[k].x, r.a[k].y, r.b[k].y);
            d[1] = make_float4(r.a[k].z, r.b[k].z, r.a[k].w, r.b[k].w);
        }
    }
}

// emit4(q, x1[4], v1[4], x2[4], v2[4]) as in unmix_pair
template <int IO, bool PADDED = false, class Emit4>
__device__ __forceinline__ void unmix_pair_regs(const cf* __restrict__ Z, const PairRegs<IO>& r, int n, int tid, int T,
                                                Emit4 emit4) {
    const int nq = n >> 2;
#pragma unroll
    for (int k = 0; k < IO; ++k) {
        const int q = tid + k * T;
        if (q < nq) {
            const f4a* s = reinterpret_cast<const f4a*>(Z + (PADDED ? 4 * q + 2 * (q >> 2) : 4 * q));
            const float4 z0 = s[0], z1 = s[1];
            const float xa[4] = {r.a[k].x, r.a[k].y, r.a[k].z, r.a[k].w}, xb[4] = {r.b[k].x, r.b[k].y, r.b[k].z, r.b[k].w};
            const float re[4] = {z0.x, z0.z, z1.x, z1.z}, im[4] = {z0.y, z0.w, z1.y, z1.w};
            float v1[4], v2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v1[e] = im[e] - xb[e]; v2[e] = xa[e] - re[e]; }
            emit4(q, xa, v1, xb, v2);
        }
    }
}

// |complex64| for the envelope: v_sqrt_f32 (1 ulp; the correctly rounded sqrtf costs ~10 instructions per sample, a
// fifth of these kernels' VALU work) -- the envelope's tolerance is 1e-5 absolute
__device__ __forceinline__ float envelope(float re, float im) { return __builtin_amdgcn_sqrtf(fmaf(re, re, im * im)); }

}  // namespace stof_io
