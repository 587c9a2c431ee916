// Device-resident compile-time twiddle tables for fft_small.h's analytic_ct<N, T> (one copy per translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include "fft_small.h"

namespace stof_ct {

template <int N> struct CtTwiddles {
    static constexpr int K = stof_fft::ct_plan_for(N).table;        // entries: w_N^t, t < K
    static __device__ const stof_fft::TwTable<K> table;
};
template <int N> __device__ constexpr stof_fft::TwTable<CtTwiddles<N>::K> CtTwiddles<N>::table =
    stof_fft::make_tw_table<N, CtTwiddles<N>::K>();

// copy the table into LDS (all TT threads of the work-group; the caller synchronises); returns its padded entry count
template <int N>
__device__ __forceinline__ void stage_twiddles(float2* __restrict__ lds, int tid, int TT) {
    const float2* src = reinterpret_cast<const float2*>(CtTwiddles<N>::table.w);
    for (int i = tid; i < CtTwiddles<N>::K; i += TT) lds[i] = src[i];
}
template <int N> constexpr int twiddle_lds_entries() { return (CtTwiddles<N>::K + 1) / 2 * 2; }     // keeps 16-byte alignment

// the two-level form (CtOpt::TW2): TA[64] followed by TB[ceil(K / 64)]
template <int N> struct CtTwiddles2 {
    static constexpr int K = stof_fft::ct_tw2_entries<N>();
    static __device__ const stof_fft::TwTable<K> table;
};
template <int N> __device__ constexpr stof_fft::TwTable<CtTwiddles2<N>::K> CtTwiddles2<N>::table = stof_fft::make_tw_table2<N>();
template <int N>
__device__ __forceinline__ void stage_twiddles2(float2* __restrict__ lds, int tid, int TT) {
    const float2* src = reinterpret_cast<const float2*>(CtTwiddles2<N>::table.w);
    for (int i = tid; i < CtTwiddles2<N>::K; i += TT) lds[i] = src[i];
}
template <int N> constexpr int twiddle2_lds_entries() { return (CtTwiddles2<N>::K + 1) / 2 * 2; }

}  // namespace stof_ct
