// Training step building blocks (main.py:204-248), exact fp32.
//
// Unlike the fused inference sweep, training keeps every layer's activation in HBM (the backward
// pass needs them), in channel-last layout [N][L][C] fp32, and runs layer by layer:
//
//   conv_cl_kernel        : y = epi( bias + conv_same(x, W) )  -- forward AND data-gradient (dgrad is
//                           the same convolution with the taps flipped and channels transposed,
//                           prepared once per step by repack_weights_kernel); MFMA 32x32x2 fp32
//   conv_wgrad_cl_kernel  : dW[d][o][c] += sum_t dY[t][o] X[t+d-pad][c], db[o] += sum_t dY[t][o]
//                           (MFMA with the time axis as the reduction dimension, fp32 atomics)
//   conv1 / pool / upsample / loss / AdamW: small VALU kernels
//
// Epilogue `epi` of conv_cl_kernel:  out = act(acc + bias) (+ residual),  or for the backward pass
// out = (acc (+ residual)) * act'(saved), where act' is read off the sign of the saved activation.
#include <hip/hip_runtime.h>
#include "stof_common.h"
#include "stof_hip_util.h"

using namespace stof;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int TROWF = 68;                 // LDS row stride (floats) for 64-channel rows, as in the inference kernels

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ floatx16 mma8(float4 a, float4 b, floatx16 c) {
    // 8 channels: lane (i, h) holds channels 4h..4h+3 of the group, step s contracts {s, 4+s}
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// x = hi + lo in fp16 (22 significant bits): D += Ah*Bh + Ah*Bl + Al*Bh over 16 channels, fp32 accumulate
__device__ __forceinline__ floatx16 mma16x3(uint4 ah, uint4 al, uint4 bh, uint4 bl, floatx16 c) {
    union U { uint4 u; half8 h; };
    U a, b, cl, d;
    a.u = ah; b.u = bh; cl.u = al; d.u = bl;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, d.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl.h, b.h, c, 0, 0, 0);
    return c;
}
// 4 fp32 values -> (hi, lo) fp16 quads packed as two 8-byte words
__device__ __forceinline__ void split4(float4 v, uint2& hi, uint2& lo) {
    const float4v f = {v.x, v.y, v.z, v.w};
    const half4v h = __builtin_convertvector(f, half4v);
    const float4v d = f - __builtin_convertvector(h, float4v);
    const half4v l = __builtin_convertvector(d, half4v);
    union { half4v h; uint2 u; } a, b;
    a.h = h; b.h = l;
    hi = a.u; lo = b.u;
}

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };

struct ConvParams {
    const float* x;        // [N][L][cin]
    const float* w;        // [K][cout][cin]  (tap-major; for dgrad: flipped taps, transposed channels)
    const float* bias;     // [cout] or nullptr
    const float* residual; // [N][L][cout] or nullptr : added after the activation (forward) / before the mask (backward)
    const float* saved;    // [N][L][cout] or nullptr : backward only, out *= act'(saved)
    float* y;              // [N][L][cout]
    int N, L, cin, cout, K, act, tiles_per_wf, total_tiles;
    // stream mode (period > 0): the rows are one sequence of N' blocks of `period` rows of which the first `valid_len`
    // map to rows of a dense [N'][valid_len][C] tensor and the rest are zero gaps (>= K/2 rows, so blocks do not see
    // each other); used by the inference path for the SemiGlobalBlock expand conv on the pooled grid
    int period, valid_len;
    const int* run_if;     // optional device flag: the launch does nothing while *run_if == 0 (fp32 re-run of the inference
                           // path's 'auto' precision mode, decided on the device without a host sync)
};

// One work-group: CT = 128 time rows x 64 output channels; loops over 64-wide input-channel blocks and
// taps.  Wave (mi, ni): output-channel half mi, time rows 64ni..64ni+63 as two 32-row MFMA tiles that share
// the weight fragments.  The weight tile of the next (block, tap) step is fetched from L2 into registers
// while the MFMAs of the current one run, and lands in the other half of a double-buffered LDS tile.
constexpr int CT = 128;
// BLOCKSUM: every 64-channel input block is summed in an accumulator of its own and folded into the total afterwards
// (shorter fp32 chains; used for the 2560-term reduction of the inference expand conv, where MFMA time is not the limit)
template <int PREC, bool BLOCKSUM = false>
__global__ __launch_bounds__(256) void conv_cl_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float xs[(CT + 8) * TROWF];
    __shared__ __attribute__((aligned(16))) float ws[2][64 * TROWF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mi = wave & 1, ni = wave >> 1, ln = lane & 31, lh = lane >> 5;
    const int o0 = blockIdx.y * 64;
    const int K = p.K, pad = K >> 1;
    const int rows = CT + K - 1;
    const int ncb = (p.cin + 63) >> 6;
    const int nsteps = ncb * K;
    if (p.run_if != nullptr && *p.run_if == 0) return;

    float4 wreg[4];
    auto wfetch = [&](int s) {
        const int c0 = (s / K) << 6, d = s - (s / K) * K;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, o = i >> 4, q = i & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PREC == STOF_PREC_F16X3) {
                // split layout [K][cout][cin_pad/64][64 hi | 64 lo halfs]: a 256-byte row per (tap, o, block) -- pure copy
                if (o0 + o < p.cout)
                    v = ld4(p.w + (((size_t)d * p.cout + o0 + o) * ncb + (c0 >> 6)) * 64 + 4 * q);
            } else {
                const int c = c0 + 4 * q;
                if (o0 + o < p.cout) {
                    const float* src = p.w + ((size_t)d * p.cout + o0 + o) * p.cin + c;
                    if (c + 3 < p.cin) v = ld4(src);
                    else {
                        if (c < p.cin) v.x = src[0];
                        if (c + 1 < p.cin) v.y = src[1];
                        if (c + 2 < p.cin) v.z = src[2];
                    }
                }
            }
            wreg[u] = v;
        }
    };
    // activation tile rows t0-pad .. t0+CT-1+pad of channel block cb (zero outside the waveform / channel range):
    // fetched HBM -> registers one (tile, block) ahead, so the latency hides behind the MFMAs of the current one
    constexpr int NXR = ((CT + 8) * 16 + 255) / 256;
    float4 xreg[NXR];
    auto xfetch = [&](int tile, int cb) {
        const int n = tile / p.tiles_per_wf;
        const int t0 = (tile - n * p.tiles_per_wf) * CT;
#pragma unroll
        for (int u = 0; u < NXR; ++u) {
            const int i = tid + 256 * u, r = i >> 4, q = i & 15;
            const int t = t0 - pad + r, c = (cb << 6) + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            size_t rowidx = (size_t)n * p.L + t;
            bool ok = r < rows && t >= 0 && t < p.L;
            if (p.period > 0 && ok) {
                const int nn = t / p.period, pp = t - nn * p.period;
                ok = pp < p.valid_len;
                rowidx = (size_t)nn * p.valid_len + pp;
            }
            if (ok) {
                const float* src = p.x + rowidx * p.cin + c;
                if (c + 3 < p.cin) v = ld4(src);
                else {
                    if (c < p.cin) v.x = src[0];
                    if (c + 1 < p.cin) v.y = src[1];
                    if (c + 2 < p.cin) v.z = src[2];
                }
            }
            xreg[u] = v;
        }
    };
    if ((int)blockIdx.x < p.total_tiles) {
        wfetch(0);
        xfetch(blockIdx.x, 0);
    }
    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
    const int n = tile / p.tiles_per_wf;
    const int t0 = (tile - n * p.tiles_per_wf) * CT;
    const bool more = tile + (int)gridDim.x < p.total_tiles;
    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    floatx16 tot[2];
    if constexpr (BLOCKSUM) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) tot[j][e] = 0.f;
    }
    for (int s = 0; s < nsteps; ++s) {
        const int cb = s / K, d = s - cb * K;
        if (d == 0) {
            if constexpr (BLOCKSUM) {
                if (cb > 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) { tot[j][e] += acc[j][e]; acc[j][e] = 0.f; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NXR; ++u) {
                const int i = tid + 256 * u, r = i >> 4, q = i & 15;
                if (r >= CT + 8) continue;
                if constexpr (PREC == STOF_PREC_F16X3) {
                    uint2 hi, lo;
                    split4(xreg[u], hi, lo);
                    char* row = reinterpret_cast<char*>(xs + r * TROWF);
                    *reinterpret_cast<uint2*>(row + 8 * q) = hi;
                    *reinterpret_cast<uint2*>(row + 128 + 8 * q) = lo;
                } else {
                    *reinterpret_cast<float4*>(xs + r * TROWF + 4 * q) = xreg[u];
                }
            }
            if (cb + 1 < ncb) xfetch(tile, cb + 1);
            else if (more) xfetch(tile + gridDim.x, 0);
        }
        float* wb = ws[s & 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u;
            *reinterpret_cast<float4*>(wb + (i >> 4) * TROWF + 4 * (i & 15)) = wreg[u];
        }
        __syncthreads();
        if (s + 1 < nsteps) wfetch(s + 1);
        else if (more) wfetch(0);
        const float* arow = wb + (32 * mi + ln) * TROWF + 4 * lh;
        const float* brow = xs + (64 * ni + ln + d) * TROWF + 4 * lh;
        if constexpr (PREC == STOF_PREC_F16X3) {
            const char* ar = reinterpret_cast<const char*>(wb + (32 * mi + ln) * TROWF) + 16 * lh;
            const char* br = reinterpret_cast<const char*>(xs + (64 * ni + ln + d) * TROWF) + 16 * lh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 ah = *reinterpret_cast<const uint4*>(ar + 32 * q), al = *reinterpret_cast<const uint4*>(ar + 128 + 32 * q);
                acc[0] = mma16x3(ah, al, *reinterpret_cast<const uint4*>(br + 32 * q),
                                 *reinterpret_cast<const uint4*>(br + 128 + 32 * q), acc[0]);
                acc[1] = mma16x3(ah, al, *reinterpret_cast<const uint4*>(br + 32 * TROWF * 4 + 32 * q),
                                 *reinterpret_cast<const uint4*>(br + 32 * TROWF * 4 + 128 + 32 * q), acc[1]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 a = ld4(arow + 8 * q);
                acc[0] = mma8(a, ld4(brow + 8 * q), acc[0]);
                acc[1] = mma8(a, ld4(brow + 32 * TROWF + 8 * q), acc[1]);
            }
        }
    }
    if constexpr (BLOCKSUM) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] += tot[j][e];
    }
    // epilogue: lane (ln, lh) holds time row t0 + 64ni + 32j + ln, channels o0 + 32mi + 8gg + 4lh + e
    const bool vec = (p.cout & 3) == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = t0 + 64 * ni + 32 * j + ln;
        if (t >= p.L) continue;
        size_t rowoff = ((size_t)n * p.L + t) * p.cout;
        if (p.period > 0) {
            const int nn = t / p.period, pp = t - nn * p.period;
            if (pp >= p.valid_len) continue;
            rowoff = ((size_t)nn * p.valid_len + pp) * p.cout;
        }
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int o = o0 + 32 * mi + 8 * gg + 4 * lh;
            if (o >= p.cout) continue;
            const bool v4 = vec && o + 3 < p.cout;
            float v[4], bi[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {1.f, 1.f, 1.f, 1.f};
            if (v4) {
                if (p.bias) { const float4 q = ld4(p.bias + o); bi[0] = q.x; bi[1] = q.y; bi[2] = q.z; bi[3] = q.w; }
                if (p.residual) { const float4 q = ld4(p.residual + rowoff + o); rs[0] = q.x; rs[1] = q.y; rs[2] = q.z; rs[3] = q.w; }
                if (p.saved) { const float4 q = ld4(p.saved + rowoff + o); sv[0] = q.x; sv[1] = q.y; sv[2] = q.z; sv[3] = q.w; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (o + e >= p.cout) continue;
                    if (p.bias) bi[e] = p.bias[o + e];
                    if (p.residual) rs[e] = p.residual[rowoff + o + e];
                    if (p.saved) sv[e] = p.saved[rowoff + o + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[j][4 * gg + e] + bi[e];
                if (p.saved == nullptr) {                       // forward
                    if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
                    else if (p.act == ACT_LRELU) x = x > 0.f ? x : 0.01f * x;
                    x += rs[e];
                } else {                                        // backward: (grad + residual grad) * act'(saved)
                    x += rs[e];
                    if (p.act == ACT_RELU) x = sv[e] > 0.f ? x : 0.f;
                    else if (p.act == ACT_LRELU) x = sv[e] > 0.f ? x : 0.01f * x;
                }
                v[e] = x;
            }
            if (v4) *reinterpret_cast<float4*>(p.y + rowoff + o) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (o + e < p.cout) p.y[rowoff + o + e] = v[e];
            }
        }
    }
    }   // persistent tile loop
}

// ----------------------------------------------------------------------------------------------------------------
// Split-fp16 channel-last convolution on v_mfma_f32_16x16x32_f16 for channel counts that are multiples of 64 (the twelve
// 64 -> 64 data-gradient convolutions of a training step, the 64 <-> 512 SemiGlobalBlock ones).  Same contract as
// conv_cl_kernel; what differs is the machinery, which is the inference sweep's:
//   * weights in MFMA-fragment order (repack_frag16_kernel), fetched L2 -> registers two chunks ahead -- no LDS staging
//     of weights and therefore no work-group barrier per (tap, channel block) step, only one per 64-channel input block
//   * the 16x16x32 shape (about 12 % more flops per watt than 32x32x16 on this part, tools/micro/mfma_shape_lds.hip)
//   * activation rows of 288 bytes: the operand reads (lane = (time row, k-group)) are bank-conflict free
//   * the output-channel order inside a 32-channel block is permuted (body16_out_channel) so that a lane ends up with
//     8 consecutive channels of its row: 32-byte loads / stores of bias, residual, saved activation and output
// Work-group tile: CT16 = 128 rows x 64 output channels; wave (mi, ni): channels 32 mi .., rows 64 ni .. = 2 M-tiles x 4
// N-tiles; chunk = (tap, 32-channel half) of one 64-channel input block: 4 weight fragments, 8 activation fragments, 24 MFMAs.
// ----------------------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int CT16 = 128;
constexpr int ROWB16 = 288;

__device__ __forceinline__ half8 as_h8(uint4 v) {
    union { uint4 u; half8 h; } c;
    c.u = v;
    return c.h;
}

template <int K>
__global__ __launch_bounds__(256, 2) void conv_cl16_kernel(const ConvParams p) {
    constexpr int PAD = K / 2, ROWS = CT16 + K - 1, NCH = 2 * K;          // chunks per input block
    __shared__ __attribute__((aligned(16))) char xs[(CT16 + 8) * ROWB16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave & 1, ni = wave >> 1, i16 = lane & 15, q4 = lane >> 4;
    const int ob = blockIdx.y, o0 = ob * 64;
    const int ncb = p.cin >> 6;
    const int nchunk = ncb * NCH;                                          // chunks per tile
    // fragment image: [ob][cb][tap][hh][frag 4][mi 2][lane 64][8 halves]; a wave's fragment f of chunk c:
    const uint4* const wbase = reinterpret_cast<const uint4*>(p.w) + (size_t)ob * nchunk * 512 + mi * 64 + lane;
    auto wload = [&](int c, int f) -> uint4 { return wbase[((size_t)c * 4 + f) * 128]; };
    // activation tile of one 64-channel block, HBM -> registers one (tile, block) ahead
    constexpr int NXR = ((CT16 + 8) * 16 + 255) / 256;
    float4 xreg[NXR];
    auto xfetch = [&](int tile, int cb) {
        const int n = tile / p.tiles_per_wf;
        const int t0 = (tile - n * p.tiles_per_wf) * CT16;
#pragma unroll
        for (int u = 0; u < NXR; ++u) {
            const int i = tid + 256 * u, r = i >> 4, q = i & 15;
            const int t = t0 - PAD + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < ROWS && t >= 0 && t < p.L) v = ld4(p.x + ((size_t)n * p.L + t) * p.cin + (cb << 6) + 4 * q);
            xreg[u] = v;
        }
    };
    uint4 wf[2][4];
    if ((int)blockIdx.x < p.total_tiles) {
        xfetch(blockIdx.x, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f) { wf[0][f] = wload(0, f); wf[1][f] = wload(1, f); }
    }
    auto mfma16 = [](const uint4& a, const uint4& b, floatx4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(a), as_h8(b), c, 0, 0, 0);
    };
    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        const int n = tile / p.tiles_per_wf;
        const int t0 = (tile - n * p.tiles_per_wf) * CT16;
        const bool more = tile + (int)gridDim.x < p.total_tiles;
        floatx4 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m][k][e] = 0.f;
        int c = 0;                                                         // chunk index within the tile
        for (int cb = 0; cb < ncb; ++cb) {
            __syncthreads();                                               // the previous block's fragments have been read
#pragma unroll
            for (int u = 0; u < NXR; ++u) {
                const int i = tid + 256 * u, r = i >> 4, q = i & 15;
                if (r >= CT16 + 8) continue;
                uint2 hi, lo;
                split4(xreg[u], hi, lo);
                char* row = xs + r * ROWB16;
                *reinterpret_cast<uint2*>(row + 8 * q) = hi;
                *reinterpret_cast<uint2*>(row + 128 + 8 * q) = lo;
            }
            if (cb + 1 < ncb) xfetch(tile, cb + 1);
            else if (more) xfetch(tile + gridDim.x, 0);
            __syncthreads();
            const char* const bbase = xs + (64 * ni + i16) * ROWB16 + 16 * q4;
            auto bload = [&](uint4 (&b)[4][2], int cc) {                   // cc = 2 tap + half
                const char* row = bbase + (cc >> 1) * ROWB16 + 64 * (cc & 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) { b[k][0] = *reinterpret_cast<const uint4*>(row + 16 * k * ROWB16);
                                              b[k][1] = *reinterpret_cast<const uint4*>(row + 16 * k * ROWB16 + 128); }
            };
            auto do_chunk = [&](uint4 (&w)[4], uint4 (&bcur)[4][2], uint4 (&bnext)[4][2], int cc) {
                int c2 = c + 2;
                if (c2 >= nchunk) c2 -= nchunk;                            // wraps to the next tile's first chunks (same weights)
                if (cc + 1 < NCH) bload(bnext, cc + 1);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        acc[m][k] = mfma16(w[2 * m], bcur[k][0], acc[m][k]);
                        acc[m][k + 1] = mfma16(w[2 * m], bcur[k + 1][0], acc[m][k + 1]);
                        acc[m][k] = mfma16(w[2 * m], bcur[k][1], acc[m][k]);
                        acc[m][k + 1] = mfma16(w[2 * m], bcur[k + 1][1], acc[m][k + 1]);
                        acc[m][k] = mfma16(w[2 * m + 1], bcur[k][0], acc[m][k]);
                        acc[m][k + 1] = mfma16(w[2 * m + 1], bcur[k + 1][0], acc[m][k + 1]);
                    }
                    w[2 * m] = wload(c2, 2 * m);
                    w[2 * m + 1] = wload(c2, 2 * m + 1);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
                ++c;
            };
            uint4 bf0[4][2], bf1[4][2];
            bload(bf0, 0);
#pragma unroll
            for (int cc = 0; cc < NCH; cc += 2) {
                do_chunk(wf[0], bf0, bf1, cc);
                do_chunk(wf[1], bf1, bf0, cc + 1);
            }
        }
        // ---- epilogue: lane (i16, q4) holds channels o0 + 32 mi + 8 q4 + 4 m + e of rows t0 + 64 ni + 16 k + i16
        const int och = o0 + 32 * mi + 8 * q4;
        float bi[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const float4 b0 = ld4(p.bias + och), b1 = ld4(p.bias + och + 4);
            bi[0] = b0.x; bi[1] = b0.y; bi[2] = b0.z; bi[3] = b0.w; bi[4] = b1.x; bi[5] = b1.y; bi[6] = b1.z; bi[7] = b1.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = t0 + 64 * ni + 16 * k + i16;
            if (t >= p.L) continue;
            const size_t off = ((size_t)n * p.L + t) * p.cout + och;
            float v[8], rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sv[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (p.residual) {
                const float4 a = ld4(p.residual + off), b = ld4(p.residual + off + 4);
                rs[0] = a.x; rs[1] = a.y; rs[2] = a.z; rs[3] = a.w; rs[4] = b.x; rs[5] = b.y; rs[6] = b.z; rs[7] = b.w;
            }
            if (p.saved) {
                const float4 a = ld4(p.saved + off), b = ld4(p.saved + off + 4);
                sv[0] = a.x; sv[1] = a.y; sv[2] = a.z; sv[3] = a.w; sv[4] = b.x; sv[5] = b.y; sv[6] = b.z; sv[7] = b.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = acc[e >> 2][k][e & 3] + bi[e];
                if (p.saved == nullptr) {                       // forward
                    if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
                    else if (p.act == ACT_LRELU) x = x > 0.f ? x : 0.01f * x;
                    x += rs[e];
                } else {                                        // backward: (grad + residual grad) * act'(saved)
                    x += rs[e];
                    if (p.act == ACT_RELU) x = sv[e] > 0.f ? x : 0.f;
                    else if (p.act == ACT_LRELU) x = sv[e] > 0.f ? x : 0.01f * x;
                }
                v[e] = x;
            }
            *reinterpret_cast<float4*>(p.y + off) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(p.y + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// operand image of conv_cl16_kernel: W'[d][a][b] (a = output, b = input index of the convolution that runs: forward
// W'[d][o][c] = w[o][c][d]; data gradient W'[d][c][o] = w[o][c][K-1-d]) as fp16 hi | lo MFMA fragments
// [a/64][b/64][d][half][frag = 2 m + part][block][lane = (i, q)][8]: a = 64 (a/64) + body16_out_channel(block, m, i),
// b = 64 (b/64) + 32 half + 8 q + e.  One thread per fp16 element.
__global__ void repack_frag16_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int cout, int cin, int K,
                                     int transpose_flip) {
    const int A = transpose_flip ? cin : cout, B = transpose_flip ? cout : cin;
    const long long total = (long long)K * A * B * 2;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), blk = (int)((i >> 9) & 1), frag = (int)((i >> 10) & 3);
    long long c = i >> 12;                                     // chunk index: ((ab * ncb + cb) * K + d) * 2 + hh
    const int hh = (int)(c & 1);
    c >>= 1;
    const int d = (int)(c % K);
    c /= K;
    const int ncb = B >> 6;
    const int cb = (int)(c % ncb), ab = (int)(c / ncb);
    const int m = frag >> 1, part = frag & 1, i16 = lane & 15, q = lane >> 4;
    const int a = 64 * ab + body16_out_channel(blk, m, i16), b = 64 * cb + 32 * hh + 8 * q + e;
    const float v = transpose_flip ? w[((size_t)b * cin + a) * K + (K - 1 - d)] : w[((size_t)a * cin + b) * K + d];
    const _Float16 h = (_Float16)v;
    out[i] = part == 0 ? h : (_Float16)(v - (float)h);
}

__host__ __device__ inline bool conv_cl16_ok(int cin, int cout, int K, int precision) {
    return precision == STOF_PREC_F16X3 && (cin & 63) == 0 && (cout & 63) == 0 && (K == 3 || K == 5 || K == 7);
}

// w_out[d][a][b] = w_in[b][a][K-1-d]  (torch layout (cout, cin, K) -> tap-major [K][cin][cout] flipped: dgrad)
// or w_out[d][o][c] = w_in[o][c][d]   (forward)
__global__ void repack_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int K,
                                      int transpose_flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cout * cin * K) return;
    if (!transpose_flip) {
        const int c = i % cin, o = (i / cin) % cout, d = i / (cin * cout);
        out[i] = w[((size_t)o * cin + c) * K + d];
    } else {
        const int o = i % cout, c = (i / cout) % cin, d = i / (cin * cout);      // out[d][c][o]
        out[i] = w[((size_t)o * cin + c) * K + (K - 1 - d)];
    }
}

// Same, as the operand image of the f16x3 mode: [K][A][B_pad/64][64 hi | 64 lo] fp16 (B padded with zeros to 64)
__global__ void repack_weights_split_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int cout, int cin, int K,
                                            int transpose_flip) {
    const int A = transpose_flip ? cin : cout, B = transpose_flip ? cout : cin;
    const int Bp = (B + 63) & ~63;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * A * Bp) return;
    const int b = i % Bp, a = (i / Bp) % A, d = i / (Bp * A);
    float v = 0.f;
    if (b < B) v = transpose_flip ? w[((size_t)b * cin + a) * K + (K - 1 - d)] : w[((size_t)a * cin + b) * K + d];
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const size_t row = ((size_t)d * A + a) * (Bp >> 6) + (b >> 6);
    out[row * 128 + (b & 63)] = h;
    out[row * 128 + 64 + (b & 63)] = l;
}

struct WgradParams {
    const float* x;    // [N][L][cin]
    const float* dy;   // [N][L][cout]
    float* part;       // workspace: [G][K][cout_pad][cin_pad] partial weight gradients
    float* dbpart;     // workspace: [G][cout_pad] partial bias gradients
    int N, L, cin, cout, K, tiles_per_wf, total_tiles, cin_pad, cout_pad;
};

// Persistent work-groups: blockIdx.x = g walks the 128-row tiles g, g+G, ... of the whole batch and keeps
// the (64 output channels) x (64 input channels) x K partial weight gradient in accumulator registers;
// each wave owns one 32x32 (o, c) quadrant and all K taps.  Time is the MFMA reduction axis (2 rows per
// MFMA).  One partial per work-group goes to the workspace; wgrad_reduce_kernel sums the G partials in a
// fixed order, so gradients are bitwise reproducible (no float atomics).
constexpr int WG_ROWS = 128;
template <int KMAX>
__global__ __launch_bounds__(256) void conv_wgrad_cl_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float dys[WG_ROWS * TROWF];
    __shared__ __attribute__((aligned(16))) float xs[(WG_ROWS + KMAX - 1) * TROWF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mi = wave & 1, ni = wave >> 1, ln = lane & 31, lh = lane >> 5;
    const int o0 = blockIdx.y * 64, c0 = blockIdx.z * 64;
    const int K = p.K, pad = K >> 1;
    floatx16 acc[KMAX];
#pragma unroll
    for (int d = 0; d < KMAX; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
    float dbs = 0.f;
    float4 dbq = make_float4(0.f, 0.f, 0.f, 0.f);
    // A[i = o][k = row parity], B[k][j = c]:  D[o][c] += dY[t][o] * X[t + d - pad][c]
    const float* ap = dys + lh * TROWF + 32 * mi + ln;
    const float* bp = xs + lh * TROWF + 32 * ni + ln;
    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        const int n = tile / p.tiles_per_wf;
        const int t0 = (tile - n * p.tiles_per_wf) * WG_ROWS;
        __syncthreads();
        for (int i = tid; i < WG_ROWS * 16; i += 256) {
            const int r = i >> 4, q = i & 15;
            const int t = t0 + r, o = o0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < p.L) {
                const float* src = p.dy + ((size_t)n * p.L + t) * p.cout + o;
                if (o + 3 < p.cout) v = ld4(src);
                else {
                    if (o < p.cout) v.x = src[0];
                    if (o + 1 < p.cout) v.y = src[1];
                    if (o + 2 < p.cout) v.z = src[2];
                }
            }
            dbq.x += v.x; dbq.y += v.y; dbq.z += v.z; dbq.w += v.w;        // bias gradient: column sums
            *reinterpret_cast<float4*>(dys + r * TROWF + 4 * q) = v;
        }
        for (int i = tid; i < (WG_ROWS + K - 1) * 16; i += 256) {
            const int r = i >> 4, q = i & 15;
            const int t = t0 - pad + r, c = c0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < p.L) {
                const float* src = p.x + ((size_t)n * p.L + t) * p.cin + c;
                if (c + 3 < p.cin) v = ld4(src);
                else {
                    if (c < p.cin) v.x = src[0];
                    if (c + 1 < p.cin) v.y = src[1];
                    if (c + 2 < p.cin) v.z = src[2];
                }
            }
            *reinterpret_cast<float4*>(xs + r * TROWF + 4 * q) = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int r = 0; r < WG_ROWS; r += 2) {
            const float a = ap[r * TROWF];
#pragma unroll
            for (int d = 0; d < KMAX; ++d)
                if (d < K) acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[(r + d) * TROWF], acc[d], 0, 0, 0);
        }
    }
    // bias gradient: thread (row lane tid >> 4, column quad tid & 15) holds the sums of its rows; fold the 16 row lanes
    if (blockIdx.z == 0) {
        __syncthreads();
        *reinterpret_cast<float4*>(xs + (tid >> 4) * 64 + 4 * (tid & 15)) = dbq;
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += xs[r * 64 + tid];
            dbs = s;
        }
    }
    // accumulator register v: row (o) = 32mi + (v&3) + 8(v>>2) + 4lh, col (c) = 32ni + ln
    float* part = p.part + (size_t)blockIdx.x * K * p.cout_pad * p.cin_pad;
    const int c = c0 + 32 * ni + ln;
#pragma unroll
    for (int d = 0; d < KMAX; ++d) {
        if (d >= K) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int o = o0 + 32 * mi + (v & 3) + 8 * (v >> 2) + 4 * lh;
            part[((size_t)d * p.cout_pad + o) * p.cin_pad + c] = acc[d][v];
        }
    }
    if (blockIdx.z == 0 && tid < 64) p.dbpart[(size_t)blockIdx.x * p.cout_pad + o0 + tid] = dbs;
}

// f16x3 variant: both operands are split into fp16 hi + lo while they are staged in LDS (rows [t][64 hi | 64 lo]),
// and the time-contiguous MFMA operand fragments come out of gfx950's transposing LDS read (ds_read_b64_tr_b16: per
// 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major), so the tap shift is just a
// row offset.  64 time rows per tile; the 320-byte row stride makes the 4 rows x 64 B of a half-wave hit 64 distinct banks.
constexpr int WG_ROWS_H = 64;
constexpr int HSTRIDE = 320;
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ uint2 tr_read(const char* p) {
    const fp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p));
    uint2 u;
    __builtin_memcpy(&u, &r, 8);
    return u;
}
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2v bits_h2w(unsigned u) { half2v h; __builtin_memcpy(&h, &u, 4); return h; }
// (gx, G) = this work-group's index among the G persistent groups of its weight block; (o0, c0) = the block; cz = its c index
// xsplit / dysplit (r4, wave-uniform): the operand is stored as SPLIT ROWS [64 x fp16 hi | 64 x fp16 lo] (the training sweeps' dump
// format, 256 bytes per row like fp32) and needs 64 channels: its rows go HBM -> registers -> LDS as they are, 16 bytes per
// lane, with no vector-pipe work -- the fp32 form converts every value to hi + lo here (four instructions per value: a quarter of
// the kernel's issue slots, and the kernel is power-limited).
// FIXK: the kernel size is KMAX (the batched launch): with a run-time K every tap sat behind its own branch, i.e. in its own basic
// block -- four LDS reads, a wait, three MFMAs, with no read of the next tap in flight.
template <int KMAX, bool FIXK = false>
__device__ __forceinline__ void conv_wgrad_f16x3_body(const WgradParams& p, const int gx, const int G, const int o0, const int c0, const int cz,
                                                      const bool xsplit = false, const bool dysplit = false) {
    __shared__ __attribute__((aligned(16))) char dys[WG_ROWS_H * HSTRIDE];
    __shared__ __attribute__((aligned(16))) char xs[(WG_ROWS_H + 8) * HSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mi = wave & 1, ni = wave >> 1, ln = lane & 31, lh = lane >> 5;
    const int K = FIXK ? KMAX : p.K, pad = K >> 1;
    floatx16 acc[KMAX];
#pragma unroll
    for (int d = 0; d < KMAX; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
    float dbs = 0.f;
    // lane's address inside a 4-row x 16-column block: row q, columns 4pq..4pq+3 of column block cb; k-group kg
    const int q = (lane & 15) >> 2, pq = lane & 3, cb = (lane >> 4) & 1, kg = lane >> 5;
    const char* const ap = dys + (8 * kg + q) * HSTRIDE + (32 * mi + 16 * cb + 4 * pq) * 2;
    const char* const bp = xs + (8 * kg + q) * HSTRIDE + (32 * ni + 16 * cb + 4 * pq) * 2;
    // The next tile's rows travel HBM -> registers while the MFMAs of the current tile run.
    constexpr int NDY = WG_ROWS_H * 16 / 256, NX = ((WG_ROWS_H + 8) * 16 + 255) / 256;
    float4 rdy[NDY], rx[NX];
    float4 dbq = make_float4(0.f, 0.f, 0.f, 0.f);
    float dbh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // split rows: sums of the thread's piece column
    auto fetch = [&](int tile) {
        const int n = tile / p.tiles_per_wf;
        const int t0 = (tile - n * p.tiles_per_wf) * WG_ROWS_H;
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int i = tid + 256 * u, r = i >> 4, qq = i & 15;
            const int t = t0 + r, o = o0 + 4 * qq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dysplit) {
                if (t < p.L) v = ld4(p.dy + ((size_t)n * p.L + t) * 64 + 4 * qq);        // piece qq of the split row: hi 0..7 | lo 8..15
            } else if (t < p.L) {
                const float* src = p.dy + ((size_t)n * p.L + t) * p.cout + o;
                if (o + 3 < p.cout) v = ld4(src);
                else {
                    if (o < p.cout) v.x = src[0];
                    if (o + 1 < p.cout) v.y = src[1];
                    if (o + 2 < p.cout) v.z = src[2];
                }
            }
            rdy[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int i = tid + 256 * u, r = i >> 4, qq = i & 15;
            const int t = t0 - pad + r, c = c0 + 4 * qq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xsplit) {
                if (t >= 0 && t < p.L && r < WG_ROWS_H + K - 1) v = ld4(p.x + ((size_t)n * p.L + t) * 64 + 4 * qq);
            } else if (t >= 0 && t < p.L && r < WG_ROWS_H + K - 1) {
                const float* src = p.x + ((size_t)n * p.L + t) * p.cin + c;
                if (c + 3 < p.cin) v = ld4(src);
                else {
                    if (c < p.cin) v.x = src[0];
                    if (c + 1 < p.cin) v.y = src[1];
                    if (c + 2 < p.cin) v.z = src[2];
                }
            }
            rx[u] = v;
        }
    };
    if (gx < p.total_tiles) fetch(gx);
    for (int tile = gx; tile < p.total_tiles; tile += G) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int i = tid + 256 * u, r = i >> 4, qq = i & 15;
            if (dysplit) {
                // the thread's piece column qq is the same for every row: 8 channels' hi (qq < 8) or lo halves; bias gradient = their sums
                const float4 v = rdy[u];
                const unsigned w4[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const half2v hv = bits_h2w(w4[e]);
                    dbh[2 * e] += (float)hv[0];
                    dbh[2 * e + 1] += (float)hv[1];
                }
                *reinterpret_cast<float4*>(dys + r * HSTRIDE + 16 * qq) = v;
            } else {
                uint2 hi, lo;
                dbq.x += rdy[u].x; dbq.y += rdy[u].y; dbq.z += rdy[u].z; dbq.w += rdy[u].w;   // bias gradient: column sums
                split4(rdy[u], hi, lo);
                *reinterpret_cast<uint2*>(dys + r * HSTRIDE + 8 * qq) = hi;
                *reinterpret_cast<uint2*>(dys + r * HSTRIDE + 128 + 8 * qq) = lo;
            }
        }
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int i = tid + 256 * u, r = i >> 4, qq = i & 15;
            if (r < WG_ROWS_H + 8) {
                if (xsplit) {
                    *reinterpret_cast<float4*>(xs + r * HSTRIDE + 16 * qq) = rx[u];
                } else {
                    uint2 hi, lo;
                    split4(rx[u], hi, lo);
                    *reinterpret_cast<uint2*>(xs + r * HSTRIDE + 8 * qq) = hi;
                    *reinterpret_cast<uint2*>(xs + r * HSTRIDE + 128 + 8 * qq) = lo;
                }
            }
        }
        __syncthreads();
        if (tile + G < p.total_tiles) fetch(tile + G);
#pragma unroll
        for (int ks = 0; ks < WG_ROWS_H / 16; ++ks) {
            const char* a = ap + ks * 16 * HSTRIDE;
            const uint2 a0 = tr_read(a), a1 = tr_read(a + 4 * HSTRIDE);
            const uint2 l0 = tr_read(a + 128), l1 = tr_read(a + 4 * HSTRIDE + 128);
            const uint4 ah = make_uint4(a0.x, a0.y, a1.x, a1.y), al = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
            for (int d = 0; d < KMAX; ++d) {
                if (d >= K) continue;
                const char* b = bp + (ks * 16 + d) * HSTRIDE;
                const uint2 b0 = tr_read(b), b1 = tr_read(b + 4 * HSTRIDE);
                const uint2 m0 = tr_read(b + 128), m1 = tr_read(b + 4 * HSTRIDE + 128);
                acc[d] = mma16x3(ah, al, make_uint4(b0.x, b0.y, b1.x, b1.y), make_uint4(m0.x, m0.y, m1.x, m1.y), acc[d]);
            }
        }
    }
    // bias gradient: thread (row lane tid >> 4, column quad tid & 15) holds the sums of its rows; fold the 16 row lanes
    if (cz == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(xs);
        if (dysplit) {
            // row lane tid >> 4, piece column pc: channels 8 (pc & 7) .. + 7 of part pc >> 3 -> red[row lane][part][channel]
            const int pc = tid & 15;
            float* const o = red + (tid >> 4) * 128 + (pc >> 3) * 64 + 8 * (pc & 7);
            *reinterpret_cast<float4*>(o) = make_float4(dbh[0], dbh[1], dbh[2], dbh[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(dbh[4], dbh[5], dbh[6], dbh[7]);
        } else {
            *reinterpret_cast<float4*>(red + (tid >> 4) * 64 + 4 * (tid & 15)) = dbq;
        }
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            if (dysplit) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s += red[r * 128 + tid] + red[r * 128 + 64 + tid];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s += red[r * 64 + tid];
            }
            dbs = s;
        }
    }
    float* part = p.part + (size_t)gx * K * p.cout_pad * p.cin_pad;
    const int c = c0 + 32 * ni + ln;
#pragma unroll
    for (int d = 0; d < KMAX; ++d) {
        if (d >= K) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int o = o0 + 32 * mi + (v & 3) + 8 * (v >> 2) + 4 * lh;
            part[((size_t)d * p.cout_pad + o) * p.cin_pad + c] = acc[d][v];
        }
    }
    if (cz == 0 && tid < 64) p.dbpart[(size_t)gx * p.cout_pad + o0 + tid] = dbs;
}

template <int KMAX>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16x3_kernel(const WgradParams p) {
    conv_wgrad_f16x3_body<KMAX>(p, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y * 64, (int)blockIdx.z * 64, (int)blockIdx.z);
}

// r4: the weight gradients of SEVERAL 64 -> 64 layers in one launch (the eleven k7 layers of StofNet's body: eleven launches of
// 97 us each + eleven reductions of 58 MB of partials per step in r3).  blockIdx.y = layer; every layer gets G persistent groups,
// G chosen so that the whole launch is ~2 work-groups per CU: 11 x fewer partials to write and to reduce.
constexpr int WGRAD_BATCH_MAX = 12;
struct WgradBatch {
    const float* x[WGRAD_BATCH_MAX];
    const float* dy[WGRAD_BATCH_MAX];
    float* dw[WGRAD_BATCH_MAX];
    float* db[WGRAD_BATCH_MAX];
    float* part;               // [count][G][K][64][64]
    float* dbpart;             // [count][G][64]
    int N, L, K, tiles_per_wf, total_tiles, count;
    unsigned x_split, dy_split;    // bit i: operand of layer i is stored as split rows
    float out_scale;
};
template <int KMAX>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16x3_batch_kernel(const WgradBatch b) {
    const int layer = blockIdx.y, G = gridDim.x;
    WgradParams p;
    p.x = b.x[layer]; p.dy = b.dy[layer];
    p.part = b.part + (size_t)layer * G * b.K * 64 * 64;
    p.dbpart = b.dbpart + (size_t)layer * G * 64;
    p.N = b.N; p.L = b.L; p.cin = 64; p.cout = 64; p.K = b.K; p.tiles_per_wf = b.tiles_per_wf; p.total_tiles = b.total_tiles;
    p.cin_pad = 64; p.cout_pad = 64;
    conv_wgrad_f16x3_body<KMAX, true>(p, (int)blockIdx.x, G, 0, 0, 0, (b.x_split >> layer) & 1u, (b.dy_split >> layer) & 1u);
}

// dw[o][c][d] = sum_g part[g][d][o][c];  db[o] = sum_g dbpart[g][o]   (fixed summation order).
// One work-group: 64 consecutive elements x 4 interleaved slices of g, combined through LDS.
constexpr int WRED_SLICES = 16;                                   // interleaved slices of g per work-group (one wave each)
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, const float* __restrict__ dbpart,
                                                  float* __restrict__ dw, float* __restrict__ db, int G, int K,
                                                  int cout, int cin, int cout_pad, int cin_pad, float out_scale) {
    __shared__ float red[WRED_SLICES][64];
    const int per = K * cout_pad * cin_pad;
    const int e = threadIdx.x & 63, gq = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    // The launch has only ~450 work-groups of 64 elements: with four waves each and two loads per wave in flight the 58 MB
    // of partials of a layer came in at 1.8 TB/s (32 us).  Sixteen waves per group, four chains each: 16 x more on its way.
    // Slices and chains are combined in a fixed order.
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < per) {
        int g = gq;
        for (; g + 3 * WRED_SLICES < G; g += 4 * WRED_SLICES) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += part[(size_t)(g + WRED_SLICES * k) * per + i];
        }
        for (; g < G; g += WRED_SLICES) a[0] += part[(size_t)g * per + i];
    } else if (i - per < cout_pad) {
        for (int g = gq; g < G; g += WRED_SLICES) a[0] += dbpart[(size_t)g * cout_pad + (i - per)];
    }
    red[gq][e] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (gq != 0) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < WRED_SLICES; q += 4) s += (red[q][e] + red[q + 1][e]) + (red[q + 2][e] + red[q + 3][e]);
    s *= out_scale;
    if (i < per) {
        const int c = i % cin_pad, o = (i / cin_pad) % cout_pad, d = i / (cin_pad * cout_pad);
        if (o < cout && c < cin) dw[((size_t)o * cin + c) * K + d] = s;
    } else if (db != nullptr && i - per < cout) {
        db[i - per] = s;
    }
}
__global__ __launch_bounds__(64 * WRED_SLICES) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ dbpart,
                                                                        float* __restrict__ dw, float* __restrict__ db, int G, int K,
                                                                        int cout, int cin, int cout_pad, int cin_pad, float out_scale) {
    wgrad_reduce_body(part, dbpart, dw, db, G, K, cout, cin, cout_pad, cin_pad, out_scale);
}
__global__ __launch_bounds__(64 * WRED_SLICES) void wgrad_reduce_batch_kernel(const WgradBatch b, int G) {
    const int layer = blockIdx.y;
    wgrad_reduce_body(b.part + (size_t)layer * G * b.K * 64 * 64, b.dbpart + (size_t)layer * G * 64, b.dw[layer], b.db[layer], G, b.K,
                      64, 64, 64, 64, b.out_scale);
}

// ----------------------------------------------------------------------------------------------------------------
// SemiGlobalBlock backward, contract_conv weight gradient from the pool's SPARSE gradient.  The gradient of
// c = lrelu(contract_conv(a1)) behind MaxPool1d(S, S) is zero except at one time row per (waveform, window, channel)
// -- 1 of S = 80 rows -- yet stof_train_pool_bwd + stof_train_wgrad build the dense [N, L, C] gradient (a 1-GB memset at
// the benched shape) and run the dense weight-gradient kernel over it (0.59 ms of a 5.8 ms step).  Here a non-zero
// (n, w, co) at row l* = w S + arg adds s a1[n][l* + d - 2][:] to dW[co][:][d], s = gpool * lrelu'(pooled):
//   work-group = (block of 128 channels, share g of the (n, w) windows); it stages the window's S + 4 rows of a1 in LDS
//   (the next window's rows are requested before the current one is used), wave k owns the 16 channels k, k + 8, ... of
//   the block and keeps their 5 x 64 partial gradients in registers (lane = input channel): per non-zero five 256-byte
//   LDS row reads and five FMAs.  One partial per work-group goes to the workspace; sgb_wgrad_reduce_kernel sums the G
//   partials in a fixed order (bitwise reproducible, no float atomics) into the parameter's layout [C][64][5].
// ----------------------------------------------------------------------------------------------------------------
constexpr int SGBW_CH = 128, SGBW_OWN = 16;                       // channels per work-group, per wave
constexpr int SGBW_PIECES = 3, SGBW_ROWS = SGBW_PIECES * 512 / 16; // 16-byte pieces of a slab per thread; slab rows staged (>= S + 4)
constexpr int SGBW_BUF_F = SGBW_ROWS * 64 + 3 * SGBW_CH;          // floats of one staging buffer: slab | gpool | pooled | arg (a dword each)
constexpr int SGBW_NBUF = 3;
struct SgbWgradParams {
    const float* gpool;        // [N][P][C]
    const unsigned char* arg;  // [N][P][C]
    const float* pooled;       // [N][P][C]
    const float* a1;           // [N][L][64]
    float* part;               // [G][C][5][64]
    float* dbpart;             // [G][C]
    long long nwin;            // N * P
    int L, P, C, S, G;
};

// s_waitcnt vmcnt(n), lgkmcnt(0) (gfx9 encoding; expcnt unconstrained)
template <int n> __device__ __forceinline__ void wait_vm_lgkm0() { __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | ((n >> 4) << 14)); }

// A work-group walks its windows q = g, g + G, ... with the inputs of the next two windows on their way: everything it
// reads -- the S + 4 rows of a1, the 128 channels' gpool / pooled / arg -- is copied global -> LDS by global_load_lds
// (no registers, no second pass through the vector pipe) into one of three buffers, retired by a counted vmcnt and one
// raw barrier per window.  (An earlier form fetched the next window into registers one iteration ahead: 252 us -- each
// window waited out a full memory round trip, ~5 us for ~1 us of work.)
__global__ __launch_bounds__(512, 4) void sgb_contract_wgrad_kernel(const SgbWgradParams p) {      // (128 registers: two groups per CU)
    extern __shared__ __attribute__((aligned(16))) float stage[];               // [SGBW_NBUF][SGBW_BUF_F]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.y, g = blockIdx.x, nf4 = (p.S + 4) * 16;
    float acc[SGBW_OWN][5];
#pragma unroll
    for (int j = 0; j < SGBW_OWN; ++j)
#pragma unroll
        for (int d = 0; d < 5; ++d) acc[j][d] = 0.f;
    float dbv = 0.f;                                             // lane j < 16: bias gradient of owned channel j
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // every wave issues SGBW_PIECES slab pieces; waves 0-1 add gpool, 2-3 pooled, 4-5 arg (one instruction each)
    auto issue = [&](long long q, int b) {
        float* const buf = stage + (size_t)b * SGBW_BUF_F;
        const long long n = q / p.P;
        const int w = (int)(q - n * p.P);
        const float* const src = p.a1 + n * (long long)p.L * 64;
#pragma unroll
        for (int k = 0; k < SGBW_PIECES; ++k) {
            const int i = tid + 512 * k;
            int l = w * p.S - 2 + (i >> 4);                       // rows outside the waveform: a valid address, zeroed after the copy
            l = l < 0 ? 0 : (l > p.L - 1 ? p.L - 1 : l);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (long long)l * 64 + 4 * (i & 15)), (lptr_t)(buf + 4 * (i - lane)), 16, 0, 0);
        }
        const long long c0 = q * p.C + SGBW_CH * blk + 64 * (wave & 1);
        float* const gp = buf + SGBW_ROWS * 64;
        if (wave < 2) __builtin_amdgcn_global_load_lds((gptr_t)(p.gpool + c0 + lane), (lptr_t)(gp + 64 * (wave & 1)), 4, 0, 0);
        else if (wave < 4) __builtin_amdgcn_global_load_lds((gptr_t)(p.pooled + c0 + lane), (lptr_t)(gp + SGBW_CH + 64 * (wave & 1)), 4, 0, 0);
        else if (wave < 6)
            __builtin_amdgcn_global_load_lds((gptr_t)(p.arg + c0 + lane), (lptr_t)(gp + 2 * SGBW_CH + 64 * (wave & 1)), 1, 0, 0);
        // (a sub-dword copy still lands one DWORD per lane, zero-extended)
    };
    long long q = g;
    if (q < p.nwin) issue(q, 0);
    if (q + p.G < p.nwin) issue(q + p.G, 1);
    int b = 0;
    for (; q < p.nwin; q += p.G) {
        // this wave's copies of window q have landed (the next window's stay in flight); the barrier extends that to every
        // wave's copies and says that everybody is done with the buffer the copies of window q + 2 G go to
        if (q + p.G < p.nwin) { if (wave < 6) wait_vm_lgkm0<SGBW_PIECES + 1>(); else wait_vm_lgkm0<SGBW_PIECES>(); }
        else wait_vm_lgkm0<0>();
        __builtin_amdgcn_s_barrier();
        if (q + 2ll * p.G < p.nwin) issue(q + 2ll * p.G, b == 0 ? 2 : b - 1);
        float* const buf = stage + (size_t)b * SGBW_BUF_F;
        const long long n = q / p.P;
        const int w = (int)(q - n * p.P);
        if (w == 0 || w * p.S + p.S + 2 > p.L) {                  // wave-uniform: rows before / behind the waveform are zero padding
            for (int i = tid; i < nf4; i += 512) {
                const int l = w * p.S - 2 + (i >> 4);
                if (l < 0 || l >= p.L) *reinterpret_cast<float4*>(buf + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        }
        const float* const gp = buf + SGBW_ROWS * 64;
        float s_cur = 0.f;
        int p_cur = 0;
        if (lane < SGBW_OWN) {
            const int ch = wave + 8 * lane;
            const float gv = gp[ch];
            s_cur = gp[SGBW_CH + ch] > 0.f ? gv : 0.01f * gv;     // the activation at the arg-max IS the pooled value
            p_cur = reinterpret_cast<const int*>(gp + 2 * SGBW_CH)[ch];
        }
        dbv += s_cur;
#pragma unroll
        for (int j = 0; j < SGBW_OWN; ++j) {
            const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s_cur), j));      // wave-uniform
            const int pj = __builtin_amdgcn_readlane(p_cur, j);
            const float* row = buf + pj * 64 + lane;              // rows pj .. pj + 4 = l* - 2 .. l* + 2
#pragma unroll
            for (int d = 0; d < 5; ++d) acc[j][d] = fmaf(sj, row[64 * d], acc[j][d]);
        }
        b = b == SGBW_NBUF - 1 ? 0 : b + 1;
    }
    float* out = p.part + ((size_t)g * p.C + SGBW_CH * blk) * 320;
#pragma unroll
    for (int j = 0; j < SGBW_OWN; ++j)
#pragma unroll
        for (int d = 0; d < 5; ++d) out[(size_t)(wave + 8 * j) * 320 + 64 * d + lane] = acc[j][d];
    if (lane < SGBW_OWN) p.dbpart[(size_t)g * p.C + SGBW_CH * blk + wave + 8 * lane] = dbv;
}

// ----------------------------------------------------------------------------------------------------------------
// r4: the batched 64 -> 64 weight gradients when EVERY operand is stored as split rows: the tiles go HBM -> LDS by
// global_load_lds (no registers, no vector-pipe pass) into one of THREE tile buffers, two tiles ahead of the MFMAs -- the
// register-staged kernel above has one tile in flight per work-group and both work-groups of a CU end up waiting for the same
// ~2.5 us round trip behind a ~1.4 us MFMA phase (0.86 ms for the eleven layers, MFMA busy 57 %).  One work-group of EIGHT
// waves per CU: wave (mi, ni, kh) owns the 32 x 32 block (mi, ni) of every tap for the K-steps 2 kh, 2 kh + 1 of each 64-row
// tile, so the two waves of a SIMD work on the same tile; every (work-group, kh) writes one partial.
// LDS rows are 256 bytes, unpadded, so that ONE copy instruction fills four rows (64 lanes x 16 bytes, consecutive in LDS; a
// first form with the 320-byte row stride needed one 16-lane instruction per row: 134 per tile at the texture addresser's
// 16 cycles each = as long as the tile's MFMAs -- 1.42 ms).  Bank conflicts of the transposing reads (a half-wave reads 64 bytes
// of four consecutive rows) are avoided by a swizzle instead: 16-byte piece p of row R sits at slot p ^ ((R & 3) << 2); the copy
// applies it on the global side (lane = slot fetches piece slot ^ ...), the readers through four precomputed lane offsets.
// Rows outside the waveform are copied from a clamped address and zeroed after they land (first / last tile of a waveform only:
// one more barrier).
// ----------------------------------------------------------------------------------------------------------------
// One copy (64 lanes x 16 bytes -> 1,024 consecutive LDS bytes at the wave-uniform address lds_addr) as inline assembly: behind
// __builtin_amdgcn_global_load_lds the compiler puts s_waitcnt vmcnt(0) in front of the next LDS read it cannot prove disjoint --
// i.e. in front of every tile's reads, which waits for the copies of the tile two ahead as well.  Hidden from its counters, these
// copies only make the waits it inserts for its own loads more conservative; ours are the explicit vmcnt below.
__device__ __forceinline__ void async_rows4(const float* src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory", "m0");
}
constexpr int WA_ROWB = 256;
constexpr int WA_XROWS = WG_ROWS_H + 8;
constexpr int WA_BUF_BYTES = (WG_ROWS_H + WA_XROWS) * WA_ROWB;         // 34,816
// KH = 2: eight waves per work-group, one work-group per CU, three buffers (copies two tiles ahead); KH = 1: four waves (every
// wave runs all four K-steps of a tile), two work-groups per CU with two buffers each (copies one tile ahead), whose phases
// drift apart so that one group's barrier / first-read bubble is filled by the other's MFMAs.
template <int K, int KH>
__global__ __launch_bounds__(256 * KH, 3 - KH) void conv_wgrad_split_async_kernel(const WgradBatch b) {
    constexpr int NW = 4 * KH, NT = 64 * NW, NBUF = KH + 1;
    extern __shared__ __attribute__((aligned(16))) char wa_lds[];       // [WA_NBUF][dy: 64 rows | x: 72 rows][256 B, swizzled]
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave & 1, ni = (wave >> 1) & 1, kh = wave >> 2;       // (kh = 0 for KH = 1)
    const int layer = blockIdx.y, G = gridDim.x, gx = blockIdx.x;
    const float* const X = b.x[layer];
    const float* const DY = b.dy[layer];
    const int L = b.L, pad = K >> 1;
    static_assert(K - 1 <= 8, "the x part of a tile buffer holds 64 + 8 rows");
    constexpr int NI = (WG_ROWS_H + WA_XROWS) / 4;                      // copy instructions per tile: 16 of dy, 18 of x
    floatx16 acc[K];
#pragma unroll
    for (int d = 0; d < K; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
    float dbh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // transposing reads: lane (q = row in the 4-row block, pq, cb, kg) reads 8 bytes = half `pq & 1` of piece 4 m + 2 cb + (pq >> 1)
    // (m = mi for dy, ni for x; + 8: lo part) of row 8 kg + q (+ 4) + 16 ks + d; (row & 3) = (q + d) & 3
    const int q = (lane & 15) >> 2, pq = lane & 3, cb = (lane >> 4) & 1, kg = lane >> 5;
    auto lane_off = [&](int m, int j) {                                  // byte offset inside the row for swizzle (q + j) & 3, hi part
        const int piece = 4 * m + 2 * cb + (pq >> 1);
        return ((piece ^ (((q + j) & 3) << 2)) << 4) + 8 * (pq & 1);
    };
    const int arow = (8 * kg + q) * WA_ROWB;
    const int aoff = arow + lane_off(mi, 0);
    int boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = WG_ROWS_H * WA_ROWB + arow + lane_off(ni, j);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)wa_lds);
    // wave w issues copy instructions w, w + 8, ...: instruction g fills rows 4 g .. 4 g + 3 of the buffer
    const int lrow = lane >> 4, slot = lane & 15;
    auto issue = [&](int tile, int buf) {
#ifdef WA_EXP_L2
        tile = gx;                                  // timing experiment: every tile re-reads the first one (L2 hits, no HBM traffic)
#endif
        const int n = tile / b.tiles_per_wf;
        const int t0 = (tile - n * b.tiles_per_wf) * WG_ROWS_H;
        const size_t wf = (size_t)n * L;
#pragma unroll 1
        for (int g = wave; g < NI; g += NW) {
            const int R = 4 * g + lrow;
            const bool isx = g >= WG_ROWS_H / 4;
            int t = isx ? t0 - pad + (R - WG_ROWS_H) : t0 + R;
            t = t < 0 ? 0 : (t > L - 1 ? L - 1 : t);
            const int piece = slot ^ ((R & 3) << 2);
            async_rows4((isx ? X : DY) + (wf + t) * 64 + 4 * piece, lds0 + buf * WA_BUF_BYTES + g * 4 * WA_ROWB);
        }
    };
    constexpr int MINE_LO = NI / NW, REM = NI % NW;                    // waves < REM issue MINE_LO + 1 copies per tile
    int tile = gx;
    if (tile < b.total_tiles) issue(tile, 0);
    if (KH == 2 && tile + G < b.total_tiles) issue(tile + G, 1);
    int buf = 0;
    for (; tile < b.total_tiles; tile += G) {
        // this wave's copies of `tile` have landed (KH = 2: those of the next tile stay in flight); the barrier extends that to every
        // wave and says that everybody is done with the buffer the next copies go to
        if (KH == 2 && tile + G < b.total_tiles) { if (wave < REM) wait_vm_lgkm0<MINE_LO + 1>(); else wait_vm_lgkm0<MINE_LO>(); }
        else wait_vm_lgkm0<0>();
        __builtin_amdgcn_s_barrier();
        if (KH == 2) { if (tile + 2 * G < b.total_tiles) issue(tile + 2 * G, buf == 0 ? 2 : buf - 1); }
        else if (tile + G < b.total_tiles) issue(tile + G, buf ^ 1);
        char* const base = wa_lds + buf * WA_BUF_BYTES;
        const int n = tile / b.tiles_per_wf;
        const int t0 = (tile - n * b.tiles_per_wf) * WG_ROWS_H;
        if (t0 - pad < 0 || t0 + WG_ROWS_H + pad > L) {              // wave-uniform: rows before / behind the waveform are zero padding
            for (int i = tid; i < (WG_ROWS_H + WG_ROWS_H + K - 1) * 16; i += NT) {
                const int R = i >> 4;
                const int t = R >= WG_ROWS_H ? t0 - pad + (R - WG_ROWS_H) : t0 + R;
                if (t < 0 || t >= L) *reinterpret_cast<float4*>(base + R * WA_ROWB + 16 * (i & 15)) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        }
        // bias gradient: thread (row tid >> 4 (+ 32), slot tid & 15) sums the 8 halves of its slot (piece slot ^ swizzle(row):
        // the same piece for both rows, 32 being a multiple of 4)
#pragma unroll
        for (int u = 0; u < WG_ROWS_H / (NT / 16); ++u) {
            const float4 v = *reinterpret_cast<const float4*>(base + ((tid >> 4) + (NT / 16) * u) * WA_ROWB + 16 * (tid & 15));
            const unsigned w4[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half2v hv = bits_h2w(w4[e]);
                dbh[2 * e] += (float)hv[0];
                dbh[2 * e + 1] += (float)hv[1];
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 4 / KH; ++k2) {
            const int ks = (4 / KH) * kh + k2;
            const char* a = base + aoff + ks * 16 * WA_ROWB;
            const uint2 a0 = tr_read(a), a1 = tr_read(a + 4 * WA_ROWB);
            const char* al_ = base + (aoff ^ 128) + ks * 16 * WA_ROWB;
            const uint2 l0 = tr_read(al_), l1 = tr_read(al_ + 4 * WA_ROWB);
            const uint4 ah = make_uint4(a0.x, a0.y, a1.x, a1.y), al = make_uint4(l0.x, l0.y, l1.x, l1.y);
            // the taps' MFMAs part by part (hi*hi of every tap, then hi*lo, then lo*hi): consecutive MFMAs never accumulate into the
            // same registers (a chain of three on one accumulator left the pipe idle between them); per accumulator the order of the
            // three products is unchanged
            uint4 bh[K], bl[K];
#pragma unroll
            for (int d = 0; d < K; ++d) {
                const char* bb = base + boff[d & 3] + (ks * 16 + d) * WA_ROWB;
                const char* bq = base + (boff[d & 3] ^ 128) + (ks * 16 + d) * WA_ROWB;
#ifdef WA_EXP_NOB                                                        // timing experiment: no operand reads at all (results are garbage)
                (void)bb; (void)bq;
                bh[d] = make_uint4(0x3c003c00u + lane, 0x3c003c00u + d, 0x3c003c00u + ks, 0x3c003c00u);
                bl[d] = make_uint4(0x1c001c00u + lane, 0x1c001c00u + d, 0x1c001c00u + ks, 0x1c001c00u);
#else
                const uint2 b0 = tr_read(bb), b1 = tr_read(bb + 4 * WA_ROWB);
                const uint2 m0 = tr_read(bq), m1 = tr_read(bq + 4 * WA_ROWB);
                bh[d] = make_uint4(b0.x, b0.y, b1.x, b1.y);
                bl[d] = make_uint4(m0.x, m0.y, m1.x, m1.y);
#endif
            }
            union U { uint4 u; half8 h; };
            U uah, ual;
            uah.u = ah; ual.u = al;
#pragma unroll
            for (int d = 0; d < K; ++d) { U ub; ub.u = bh[d]; acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uah.h, ub.h, acc[d], 0, 0, 0); }
#pragma unroll
            for (int d = 0; d < K; ++d) { U ub; ub.u = bl[d]; acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uah.h, ub.h, acc[d], 0, 0, 0); }
#pragma unroll
            for (int d = 0; d < K; ++d) { U ub; ub.u = bh[d]; acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ual.h, ub.h, acc[d], 0, 0, 0); }
        }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
    // partial (work-group, kh): accumulator register v: row (o) = 32 mi + (v & 3) + 8 (v >> 2) + 4 lh, column (c) = 32 ni + ln
    const int Gsets = KH * G, set = KH * gx + kh;
    float* const part = b.part + ((size_t)layer * Gsets + set) * K * 64 * 64;
    const int ln = lane & 31, lh = lane >> 5, c = 32 * ni + ln;
#pragma unroll
    for (int d = 0; d < K; ++d)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int o = 32 * mi + (v & 3) + 8 * (v >> 2) + 4 * lh;
            part[((size_t)d * 64 + o) * 64 + c] = acc[d][v];
        }
    // bias gradient of the work-group -> set (gx, 0); set (gx, 1) holds zeros
    __syncthreads();
    float* const red = reinterpret_cast<float*>(wa_lds);
    {
        const int pc = (tid & 15) ^ (((tid >> 4) & 3) << 2);            // the piece behind the thread's slot
        float* const o = red + (tid >> 4) * 128 + (pc >> 3) * 64 + 8 * (pc & 7);
        *reinterpret_cast<float4*>(o) = make_float4(dbh[0], dbh[1], dbh[2], dbh[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(dbh[4], dbh[5], dbh[6], dbh[7]);
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < NT / 16; ++r) s += red[r * 128 + tid] + red[r * 128 + 64 + tid];
        float* const dbp = b.dbpart + ((size_t)layer * Gsets + KH * gx) * 64;
        dbp[tid] = s;
        if (KH == 2) dbp[64 + tid] = 0.f;
    }
}

// The four-wave form on v_mfma_f32_16x16x32_f16 (K = 32 time rows per MFMA, four 16 x 16 accumulators per tap and wave): the shape
// that holds the higher clock under the power cap (tools/micro/mfma_shape_lds.hip).  Same tiles, copies, swizzle and partials as
// conv_wgrad_split_async_kernel<K, 1>; what differs is the operand addressing of the transposing reads (lane = (i = l & 15, q =
// l >> 4): column i of the 16-column block of its M- / N-tile, rows 8 q .. 8 q + 7 of the 32-row K-step) and the accumulator layout.
template <int K>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_async16_kernel(const WgradBatch b) {
    constexpr int NW = 4, NT = 256, NBUF = 2;
    extern __shared__ __attribute__((aligned(16))) char wa_lds[];
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave & 1, ni = wave >> 1;
    const int layer = blockIdx.y, G = gridDim.x, gx = blockIdx.x;
    const float* const X = b.x[layer];
    const float* const DY = b.dy[layer];
    const int L = b.L, pad = K >> 1;
    static_assert(K - 1 <= 8, "the x part of a tile buffer holds 64 + 8 rows");
    constexpr int NI = (WG_ROWS_H + WA_XROWS) / 4;
    floatx4 acc[K][2][2];                                               // [tap][M-tile (o)][N-tile (c)]
#pragma unroll
    for (int d = 0; d < K; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[d][m][n2][e] = 0.f;
    float dbh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, q4 = lane >> 4;
    const int rsel = i16 >> 2, csel = i16 & 3;                          // the lane's row and 8-byte chunk inside the 4 x 16 block it addresses
    auto lane_off = [&](int m, int t2, int j) {                          // byte offset inside the row for swizzle (rsel + j) & 3, hi part
        const int piece = 4 * m + 2 * t2 + (csel >> 1);
        return ((piece ^ (((rsel + j) & 3) << 2)) << 4) + 8 * (csel & 1);
    };
    const int arow = (8 * q4 + rsel) * WA_ROWB;
    // (tile t2 = 1 of an operand is piece + 2: offset ^ 32, the swizzle touching bits 6 and 7 only; lo part: ^ 128)
    const int aoff0 = arow + lane_off(mi, 0, 0);
    int boff0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff0[j] = WG_ROWS_H * WA_ROWB + arow + lane_off(ni, 0, j);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)wa_lds);
    const int lrow = lane >> 4, slot = lane & 15;
    auto issue = [&](int tile, int buf) {
        const int n = tile / b.tiles_per_wf;
        const int t0 = (tile - n * b.tiles_per_wf) * WG_ROWS_H;
        const size_t wf = (size_t)n * L;
#pragma unroll 1
        for (int g = wave; g < NI; g += NW) {
            const int R = 4 * g + lrow;
            const bool isx = g >= WG_ROWS_H / 4;
            int t = isx ? t0 - pad + (R - WG_ROWS_H) : t0 + R;
            t = t < 0 ? 0 : (t > L - 1 ? L - 1 : t);
            const int piece = slot ^ ((R & 3) << 2);
            async_rows4((isx ? X : DY) + (wf + t) * 64 + 4 * piece, lds0 + buf * WA_BUF_BYTES + g * 4 * WA_ROWB);
        }
    };
    auto frag = [&](const char* a) -> half8 {                            // rows r .. r + 3 and r + 4 .. r + 7 of the lane's column
        const uint2 lo4 = tr_read(a), hi4 = tr_read(a + 4 * WA_ROWB);
        union { uint4 u; half8 h; } c;
        c.u = make_uint4(lo4.x, lo4.y, hi4.x, hi4.y);
        return c.h;
    };
    int tile = gx;
    if (tile < b.total_tiles) issue(tile, 0);
    int buf = 0;
    for (; tile < b.total_tiles; tile += G) {
        wait_vm_lgkm0<0>();
        __builtin_amdgcn_s_barrier();
        if (tile + G < b.total_tiles) issue(tile + G, buf ^ 1);
        char* const base = wa_lds + buf * WA_BUF_BYTES;
        const int n = tile / b.tiles_per_wf;
        const int t0 = (tile - n * b.tiles_per_wf) * WG_ROWS_H;
        if (t0 - pad < 0 || t0 + WG_ROWS_H + pad > L) {
            for (int i = tid; i < (WG_ROWS_H + WG_ROWS_H + K - 1) * 16; i += NT) {
                const int R = i >> 4;
                const int t = R >= WG_ROWS_H ? t0 - pad + (R - WG_ROWS_H) : t0 + R;
                if (t < 0 || t >= L) *reinterpret_cast<float4*>(base + R * WA_ROWB + 16 * (i & 15)) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int u = 0; u < WG_ROWS_H / (NT / 16); ++u) {
            const float4 v = *reinterpret_cast<const float4*>(base + ((tid >> 4) + (NT / 16) * u) * WA_ROWB + 16 * (tid & 15));
            const unsigned w4[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half2v hv = bits_h2w(w4[e]);
                dbh[2 * e] += (float)hv[0];
                dbh[2 * e + 1] += (float)hv[1];
            }
        }
#pragma unroll
        for (int kk = 0; kk < WG_ROWS_H / 32; ++kk) {
            half8 ah[2], al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = frag(base + (aoff0 ^ (32 * m)) + kk * 32 * WA_ROWB);
                al[m] = frag(base + (aoff0 ^ (32 * m) ^ 128) + kk * 32 * WA_ROWB);
            }
            // the taps as a two-stage pipeline placed by hand: tap d + 1's operand reads are issued in front of tap d's MFMAs and the
            // scheduling barrier keeps them there (left alone the compiler hoists several taps' reads and spills 25 registers)
            half8 bh[2][2], bl[2][2];                                    // [stage][N-tile]
            auto load_tap = [&](int st, int d) {
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    bh[st][n2] = frag(base + (boff0[d & 3] ^ (32 * n2)) + (kk * 32 + d) * WA_ROWB);
                    bl[st][n2] = frag(base + (boff0[d & 3] ^ (32 * n2) ^ 128) + (kk * 32 + d) * WA_ROWB);
                }
            };
            load_tap(0, 0);
#pragma unroll
            for (int d = 0; d < K; ++d) {
                const int st = d & 1;
                if (d + 1 < K) load_tap(st ^ 1, d + 1);
                // part-major over the tap's four accumulators; per accumulator hi*hi, hi*lo, lo*hi as everywhere
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) acc[d][m][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[st][n2], acc[d][m][n2], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) acc[d][m][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[st][n2], acc[d][m][n2], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) acc[d][m][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[st][n2], acc[d][m][n2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        buf ^= 1;
    }
    // partial of the work-group: D element e of lane (column i16, q4) of tile (m, n2) = (o = 32 mi + 16 m + 4 q4 + e, c = 32 ni + 16 n2 + i16)
    float* const part = b.part + ((size_t)layer * G + gx) * K * 64 * 64;
#pragma unroll
    for (int d = 0; d < K; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    part[((size_t)d * 64 + 32 * mi + 16 * m + 4 * q4 + e) * 64 + 32 * ni + 16 * n2 + i16] = acc[d][m][n2][e];
    __syncthreads();
    float* const red = reinterpret_cast<float*>(wa_lds);
    {
        const int pc = (tid & 15) ^ (((tid >> 4) & 3) << 2);
        float* const o = red + (tid >> 4) * 128 + (pc >> 3) * 64 + 8 * (pc & 7);
        *reinterpret_cast<float4*>(o) = make_float4(dbh[0], dbh[1], dbh[2], dbh[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(dbh[4], dbh[5], dbh[6], dbh[7]);
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < NT / 16; ++r) s += red[r * 128 + tid] + red[r * 128 + 64 + tid];
        b.dbpart[((size_t)layer * G + gx) * 64 + tid] = s;
    }
}

// orders a wave's LDS accesses for the compiler (the hardware executes one wave's LDS instructions in order)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ----------------------------------------------------------------------------------------------------------------
// SemiGlobalBlock backward, contract_conv data gradient from the same sparse gradient: a non-zero (n, w, co) at row
// l* = w S + arg adds s W[co][:][d] to dL/da1[n][l* + d - 2][:] for the five taps.  One WAVE per window (n, w), lane = input
// channel quad x list-entry quarter, no atomics and a fixed summation order:
//   * the wave sets a bit per (position, channel) in an LDS bit matrix for its window and for the two positions of each
//     neighbouring window that reach its rows;
//   * it walks the positions in order with FIVE accumulators in registers -- the rows a position reaches: after position
//     pi the oldest row is complete, is stored, and the accumulators shift (the incoming one starts from the residual
//     the dense convolution would have added, fetched a few rows ahead);
//   * the channels of a position come out of the bit matrix as a list (rank by mbcnt); eight at a time, their five
//     256-byte weight rows (L2-resident 655-KB table wt[co][d][ci]) and their s values are requested together: 16-byte
//     loads, a quarter of the wave per list entry, so one instruction fetches four rows.
// Replaces a 1-GB memset, stof_train_pool_bwd and a 503-GFLOP dense convolution over 98.75 % zeros (0.83 ms of a step).
// ----------------------------------------------------------------------------------------------------------------
constexpr int SGBD_WAVES = 4, SGBD_MAXC = 512, SGBD_MAXS = 88, SGBD_GROUP = 2;      // (28 KB of LDS per work-group: five per CU; 4 x GROUP list entries per step)
constexpr int SGBD_PI = SGBD_MAXS + 6;                           // positions pi = rel + 2, rel = -2 .. S + 3 (the last two only flush)
struct SgbDgradParams {
    const float* gpool;        // [N][P][C]
    const unsigned char* arg;  // [N][P][C]
    const float* pooled;       // [N][P][C]
    const float* wt;           // [C][5][64] = contract_conv.weight[co][ci][d] transposed
    const float* resid;        // [N][L][64] or nullptr
    float* out;                // [N][L][64]
    long long nwin;            // N * P
    int L, P, C, S;
};

__global__ __launch_bounds__(64 * SGBD_WAVES) void sgb_contract_dgrad_kernel(const SgbDgradParams p) {
    __shared__ unsigned long long pm_all[SGBD_WAVES][SGBD_PI][SGBD_MAXC / 64];
    __shared__ unsigned short list_all[SGBD_WAVES][2][SGBD_MAXC];          // (two lists: the next position's is built under this one's loads)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long q = (long long)blockIdx.x * SGBD_WAVES + wave;
    if (q >= p.nwin) return;                                     // (no work-group barrier below: a wave is on its own)
    unsigned long long (*pm)[SGBD_MAXC / 64] = pm_all[wave];
    const int S = p.S, L = p.L, C = p.C, NW = C / 64;
    const long long n = q / p.P;
    const int w = (int)(q - n * p.P);
    const bool last = w == p.P - 1;
    const int npi = S + 6;                                        // positions walked
    for (int i = lane; i < npi * (SGBD_MAXC / 64); i += 64) (&pm[0][0])[i] = 0ull;
    wave_lds_fence();
    // bits of the previous / this / next window (positions rel = arg + (t - 1) S; kept: -2 <= rel < S + 2); all the
    // arg-max bytes are requested before the first one is used
    {
        int av[3][SGBD_MAXC / 64];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int wn = w + t - 1, wc = wn < 0 ? 0 : (wn >= p.P ? p.P - 1 : wn);
            const unsigned char* src = p.arg + (n * p.P + wc) * (long long)C + lane;
#pragma unroll
            for (int k = 0; k < SGBD_MAXC / 64; ++k) av[t][k] = (k < NW) ? (int)src[64 * k] : 0;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int wn = w + t - 1;
#pragma unroll
            for (int k = 0; k < SGBD_MAXC / 64; ++k) {
                const int pi = av[t][k] + (t - 1) * S + 2;
                if (k < NW && wn >= 0 && wn < p.P && pi >= 0 && pi < S + 4) atomicOr(&pm[pi][k], 1ull << lane);
            }
        }
    }
    wave_lds_fence();
    // rows: r = l - w S; accumulator d at step pi holds row r = pi + d - 4.  Owned rows: 0 .. S - 1 (the last window also
    // takes the rows behind it).  Lane = (quarter sub = lane >> 4, channels 4 c4 .. 4 c4 + 3, c4 = lane & 15): a 16-byte
    // load per lane fetches FOUR weight rows per instruction, one per quarter (the texture addresser takes a wave's 64
    // addresses at the same pace whatever their width: dword loads, one row per instruction, made it the bottleneck at
    // 0.65 ms); quarter `sub` accumulates the list entries e = sub mod 4 and the quarters are added when a row is complete.
    const float* const resid = p.resid ? p.resid + n * (long long)L * 64 : nullptr;
    float* const dst = p.out + n * (long long)L * 64;
    const int l0 = w * S, r_end = last ? L - l0 : S, sub = lane >> 4, c4 = lane & 15;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto initial = [&](int r) -> float4 {                        // the residual, on quarter 0
        const int l = l0 + r;
        return (resid && sub == 0 && r >= 0 && l < L) ? *reinterpret_cast<const float4*>(resid + (long long)l * 64 + 4 * c4) : zero4;
    };
    constexpr int AHEAD = 4;
    float4 acc[5], nxt[AHEAD];
#pragma unroll
    for (int d = 0; d < 5; ++d) acc[d] = initial(d - 4);
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) nxt[a] = initial(1 + a);    // rows entering at the steps pi = 0 .. AHEAD - 1
    // the channels of position pi, ascending, into list buffer pi & 1; returns their number.  Lane l takes byte l of the
    // position's 512-bit row (channels 8 l .. 8 l + 7): an exclusive scan of the bytes' bit counts over the wave gives each
    // lane its place in the list, and a lane writes its own (rarely more than one) channels.  (A form that walked the row's
    // eight 64-bit words with readlane / mbcnt cost ~160 vector instructions per position, a quarter of the kernel.)
    auto build_list = [&](int pi) -> int {
        unsigned short* const list = list_all[wave][pi & 1];
        unsigned bits = lane < 8 * NW ? (unsigned)reinterpret_cast<const unsigned char*>(&pm[pi][0])[lane] : 0u;
        const int mine = __builtin_popcount(bits);
        int incl = mine;                                          // inclusive scan over the 64 lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        int at = incl - mine;
        while (bits) {                                            // (divergent: as many rounds as the fullest byte has channels)
            const int b = __builtin_ctz(bits);
            bits &= bits - 1u;
            list[at++] = (unsigned short)(8 * lane + b);
        }
        return __builtin_amdgcn_readlane(incl, 63);
    };
    int count = build_list(0);
    wave_lds_fence();
    for (int pi = 0; pi < npi; ++pi) {
        const int t = pi < 2 ? 0 : (pi < S + 2 ? 1 : 2);
        const unsigned short* const list = list_all[wave][pi & 1];
        int count_next = 0;
        // s = gpool * lrelu'(pooled) is read with the weight rows (keeping three windows' values in LDS would halve the waves
        // per CU)
        const int wn = w + t - 1;                                // (a neighbour that does not exist has no bits; its loads stay in range)
        const long long sbase = (n * p.P + (wn < 0 ? 0 : (wn >= p.P ? p.P - 1 : wn))) * (long long)C;
        for (int e0 = 0; e0 < count || e0 == 0; e0 += 4 * SGBD_GROUP) {
            float gx[SGBD_GROUP], px[SGBD_GROUP];
            float4 wv[SGBD_GROUP][5];
#pragma unroll
            for (int u = 0; u < SGBD_GROUP; ++u) {
                const int e = e0 + 4 * u + sub;
                const int co = e < count ? (int)list[e] : 0;      // (an entry past the list reads channel 0 and counts for nothing)
                gx[u] = p.gpool[sbase + co];
                px[u] = p.pooled[sbase + co];
                const float* wr = p.wt + (size_t)co * 320 + 4 * c4;
#pragma unroll
                for (int d = 0; d < 5; ++d) wv[u][d] = *reinterpret_cast<const float4*>(wr + 64 * d);
            }
            if (e0 == 0 && pi + 1 < npi) count_next = build_list(pi + 1);       // under the loads just issued
#pragma unroll
            for (int u = 0; u < SGBD_GROUP; ++u) {
                const float sv = px[u] > 0.f ? gx[u] : 0.01f * gx[u];     // the activation at the arg-max IS the pooled value
                const float sx = (e0 + 4 * u + sub < count) ? sv : 0.f;
#pragma unroll
                for (int d = 0; d < 5; ++d) {
                    acc[d].x = fmaf(sx, wv[u][d].x, acc[d].x); acc[d].y = fmaf(sx, wv[u][d].y, acc[d].y);
                    acc[d].z = fmaf(sx, wv[u][d].z, acc[d].z); acc[d].w = fmaf(sx, wv[u][d].w, acc[d].w);
                }
            }
        }
        count = count_next;
        wave_lds_fence();                                        // the next position's list is complete
        const int r = pi - 4;
        if (r >= 0 && r < r_end && r < S + 2) {                    // wave-uniform: the row is complete -- add the quarters, store
            float4 v = acc[0];
            v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
            v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
            if (sub == 0) *reinterpret_cast<float4*>(dst + (long long)(l0 + r) * 64 + 4 * c4) = v;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d] = acc[d + 1];
        acc[4] = nxt[0];
#pragma unroll
        for (int a = 0; a + 1 < AHEAD; ++a) nxt[a] = nxt[a + 1];
        nxt[AHEAD - 1] = initial(pi + 1 + AHEAD);
    }
    // the last window: rows no position reaches keep the residual
    if (last && sub == 0)
        for (int r = S + 2; r < r_end; ++r) *reinterpret_cast<float4*>(dst + (long long)(l0 + r) * 64 + 4 * c4) = initial(r);
}

// wt[co][d][ci] = w[co][ci][d]
__global__ __launch_bounds__(256) void sgb_wt_repack_kernel(const float* __restrict__ w, float* __restrict__ wt, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * 320) return;
    const int co = i / 320, r = i - co * 320, d = r >> 6, ci = r & 63;
    wt[i] = w[(size_t)co * 320 + ci * 5 + d];
}

// dw[co][ci][d] = scale * sum_g part[g][co][d][ci],  db[co] = scale * sum_g dbpart[g][co]   (fixed order; sixteen interleaved
// slices of g per work-group as in wgrad_reduce_kernel, for the loads in flight)
__global__ __launch_bounds__(64 * WRED_SLICES) void sgb_wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ dbpart,
                                                                            float* __restrict__ dw, float* __restrict__ db, int G, int C, float scale) {
    __shared__ float red[WRED_SLICES][64];
    const int e = threadIdx.x & 63, gq = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e, per = C * 320;            // i < per: (co, d, ci); then the C bias entries
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < per) {
        int g = gq;
        for (; g + 3 * WRED_SLICES < G; g += 4 * WRED_SLICES) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += part[(size_t)(g + WRED_SLICES * k) * per + i];
        }
        for (; g < G; g += WRED_SLICES) a[0] += part[(size_t)g * per + i];
    } else if (i - per < C) {
        for (int g = gq; g < G; g += WRED_SLICES) a[0] += dbpart[(size_t)g * C + (i - per)];
    }
    red[gq][e] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (gq != 0) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < WRED_SLICES; q += 4) s += (red[q][e] + red[q + 1][e]) + (red[q + 2][e] + red[q + 3][e]);
    s *= scale;
    if (i < per) {
        const int co = i / 320, r = i - co * 320, d = r >> 6, ci = r & 63;
        dw[(size_t)co * 320 + ci * 5 + d] = s;
    } else if (i - per < C) {
        db[i - per] = s;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Data gradient of conv_last (64 -> r channels, k3): g6[n][t][c] = sum_{o < r, d} dz[n][t - d + 1][o] w[o][c][d].  With
// r = 10 (or 4) input channels the layer kernels pad to a 64-channel block and spend 116 us at the benched shape on what
// is 2 GFLOP and 150 MB; here it runs on the vector pipe in exact fp32: a work-group takes 256 rows of one waveform,
// thread = (channel quad, row lane) keeps its 3 x R x 4 weights in registers and walks 16 rows, the dz rows come from LDS.
// ----------------------------------------------------------------------------------------------------------------
constexpr int CLD_ROWS = 256;
template <int R, bool SPLIT = false>          // SPLIT: the rows are written as split rows [64 x fp16 hi | 64 x fp16 lo]
__global__ __launch_bounds__(256) void conv_last_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                              float* __restrict__ out, int N, int L) {
    constexpr int RP = (R + 3) / 4 * 4;                           // floats per staged row
    __shared__ __attribute__((aligned(16))) float zs[(CLD_ROWS + 2) * RP];
    const int tid = threadIdx.x, q = tid & 15, rl = tid >> 4;
    const int tiles = (L + CLD_ROWS - 1) / CLD_ROWS;
    const long long n = blockIdx.x / tiles;
    const int t0 = (int)(blockIdx.x - n * tiles) * CLD_ROWS;
    float wr[3][R][4];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) wr[d][o][e] = w[((size_t)o * 64 + 4 * q + e) * 3 + d];
    for (int i = tid; i < (CLD_ROWS + 2) * RP; i += 256) {        // rows t0 - 1 .. t0 + CLD_ROWS, zero outside the waveform
        const int row = i / RP, c = i - row * RP, t = t0 - 1 + row;
        zs[i] = (c < R && t >= 0 && t < L) ? dz[(n * L + t) * R + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 2
    for (int j = 0; j < CLD_ROWS / 16; ++j) {
        const int rr = rl + 16 * j, t = t0 + rr;
        if (t >= L) break;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float* zr = zs + (rr + 2 - d) * RP;             // row t - d + 1  <->  staged row rr + 1 - d + 1
            float z[RP];
#pragma unroll
            for (int k = 0; k < RP / 4; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(zr + 4 * k);
                z[4 * k] = v.x; z[4 * k + 1] = v.y; z[4 * k + 2] = v.z; z[4 * k + 3] = v.w;
            }
#pragma unroll
            for (int o = 0; o < R; ++o)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(z[o], wr[d][o][e], acc[e]);
        }
        if constexpr (SPLIT) {
            uint2 hi, lo;
            split4(make_float4(acc[0], acc[1], acc[2], acc[3]), hi, lo);
            char* const o = reinterpret_cast<char*>(out + (n * L + t) * 64);
            *reinterpret_cast<uint2*>(o + 8 * q) = hi;
            *reinterpret_cast<uint2*>(o + 128 + 8 * q) = lo;
        } else {
            *reinterpret_cast<float4*>(out + (n * L + t) * 64 + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
}

// conv1 (1 -> 64, k9, pad 4) + ReLU, channel-last output; and its weight gradient
// A work-group takes 64 consecutive time rows of one waveform: thread = (row r = tid >> 4 (+ 16 j), channel quad q = tid & 15)
// with its 4 x 9 weights in registers; the 72-sample input window and the weights go through LDS once per group.  (One
// thread per (row, quad) reading its 36 weights and 9 samples with vector loads was bound by the texture addresser --
// a wave's 64 addresses cost the same whatever the width: 79 us for the 131 MB of output at batch 256.)
constexpr int C1_ROWS = 64;
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int N, int L) {
    __shared__ float ws[64 * 9 + 64];
    __shared__ float xs[C1_ROWS + 8];
    const int tid = threadIdx.x, q = tid & 15, r = tid >> 4;
    const int tiles = (L + C1_ROWS - 1) / C1_ROWS;
    const long long n = blockIdx.x / tiles;
    const int t0 = (int)(blockIdx.x - n * tiles) * C1_ROWS;
    for (int i = tid; i < 64 * 9 + 64; i += 256) ws[i] = i < 64 * 9 ? w[i] : b[i - 64 * 9];
    if (tid < C1_ROWS + 8) {
        const int u = t0 + tid - 4;
        xs[tid] = (u >= 0 && u < L) ? x[n * L + u] : 0.f;
    }
    __syncthreads();
    float wr[4][9], br[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        br[e] = ws[64 * 9 + 4 * q + e];
#pragma unroll
        for (int d = 0; d < 9; ++d) wr[e][d] = ws[(4 * q + e) * 9 + d];
    }
#pragma unroll
    for (int j = 0; j < C1_ROWS / 16; ++j) {
        const int rr = r + 16 * j, t = t0 + rr;
        if (t >= L) break;
        float xv[9];
#pragma unroll
        for (int d = 0; d < 9; ++d) xv[d] = xs[rr + d];
        float4 o;
        float* op = &o.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = br[e];
#pragma unroll
            for (int d = 0; d < 9; ++d) a = fmaf(wr[e][d], xv[d], a);
            op[e] = fmaxf(a, 0.f);
        }
        *reinterpret_cast<float4*>(y + (n * L + t) * 64 + 4 * q) = o;
    }
}

// dW1[ch][d] += sum_t g'[t][ch] x[t+d-4], db1[ch] += sum_t g'[t][ch], g' = g * relu'(saved conv1 output)
// A grid of <= C1_COPIES work-groups walks the row chunks in a fixed assignment (chunk c -> work-group c mod grid), every
// work-group keeps its sums in registers and writes ONE partial; the partials are added in a fixed order: no float atomics, so
// the gradient is bitwise repeatable like every other weight gradient of the step (r3 used atomics into 64 replicated copies).
constexpr int C1_COPIES = 2048;
constexpr int C1_CHUNK = 256;          // rows per chunk
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ saved,
                                                          float* __restrict__ copies, int N, int L) {
    __shared__ float red[4][64][10];
    // (part -- hence the row and its nine input samples -- is wave-uniform: made scalar so that the samples come through the
    // scalar cache; as vector loads they were nine of the eleven loads per row and the texture addresser was the limit, 149 us)
    const int tid = threadIdx.x, ch = tid & 63, part = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = N * L;                                    // N*L < 2^31 is checked by the caller
    float acc[10];
#pragma unroll
    for (int d = 0; d < 10; ++d) acc[d] = 0.f;
    for (int r0 = blockIdx.x * C1_CHUNK; r0 < total; r0 += gridDim.x * C1_CHUNK) {
        const int rend = min(r0 + C1_CHUNK, total);
        int t = (r0 + part) % L;
#pragma unroll 4
        for (int r = r0 + part; r < rend; r += 4) {
            const float* xr = x + (r - t);
            const float gl = g[(size_t)r * 64 + ch], sl = saved[(size_t)r * 64 + ch];      // (both requested before either is used)
            const float gv = sl > 0.f ? gl : 0.f;
            float xv[9];
            if (t >= 4 && t < L - 4) {                              // (scalar: all nine samples inside the waveform, no guards)
#pragma unroll
                for (int d = 0; d < 9; ++d) xv[d] = xr[t + d - 4];
            } else {
#pragma unroll
                for (int d = 0; d < 9; ++d) { const int u = t + d - 4; xv[d] = (u >= 0 && u < L) ? xr[u] : 0.f; }
            }
#pragma unroll
            for (int d = 0; d < 9; ++d) acc[d] = fmaf(gv, xv[d], acc[d]);
            acc[9] += gv;
            t += 4;
            while (t >= L) t -= L;                                  // (L < 4: more than one wrap)
        }
    }
#pragma unroll
    for (int d = 0; d < 10; ++d) red[part][ch][d] = acc[d];
    __syncthreads();
    if (part == 0) {
        float* copy = copies + blockIdx.x * 640;
#pragma unroll
        for (int d = 0; d < 10; ++d) copy[ch * 10 + d] = (red[0][ch][d] + red[1][ch][d]) + (red[2][ch][d] + red[3][ch][d]);
    }
}

// 640 sums over the partials, in a fixed order: 64 elements x 16 interleaved slices of the copies per work-group (one wave each, four
// chains per wave so that loads stay in flight), combined through LDS
__global__ __launch_bounds__(1024) void conv1_wgrad_reduce_kernel(const float* __restrict__ copies, int ncopies, float* __restrict__ dw,
                                                                  float* __restrict__ db, float out_scale) {
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;                            // (ch, d), 640 in all: 10 work-groups
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int c = q;
    for (; c + 48 < ncopies; c += 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += copies[(size_t)(c + 16 * k) * 640 + i];
    }
    for (; c < ncopies; c += 16) a[0] += copies[(size_t)c * 640 + i];
    red[q][e] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (q != 0) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k += 4) s += (red[k][e] + red[k + 1][e]) + (red[k + 2][e] + red[k + 3][e]);
    const int ch = i / 10, d = i - ch * 10;
    if (d < 9) dw[ch * 9 + d] = s * out_scale; else db[ch] = s * out_scale;
}

// Gradient with respect to the input frame (models/stofnet.py:45 differentiated: the reference's autograd provides it for free):
// dx[n][u] = out_scale * sum_d sum_ch w1[ch][d] g'[n][u + 4 - d][ch],  g' = g * relu'(saved conv1 output).  One thread per sample;
// the nine rows of g' it reads are shared with its neighbours through the caches; off the hot path (nobody trains the input).
__global__ __launch_bounds__(256) void conv1_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ saved,
                                                          const float* __restrict__ w, float* __restrict__ dx, int N, int L,
                                                          float out_scale) {
    __shared__ float ws[9][64];                                   // [tap][channel]
    for (int i = threadIdx.x; i < 576; i += 256) ws[i % 9][i / 9] = w[i];
    __syncthreads();
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)N * L) return;
    const int u = (int)(i % L);
    const long long n = i / L;
    float acc = 0.f;
    for (int d = 0; d < 9; ++d) {
        const int t = u + 4 - d;
        if (t < 0 || t >= L) continue;
        const float4* gr = reinterpret_cast<const float4*>(g + ((size_t)n * L + t) * 64);
        const float4* sr = reinterpret_cast<const float4*>(saved + ((size_t)n * L + t) * 64);
#pragma unroll 4
        for (int c4 = 0; c4 < 16; ++c4) {
            const float4 gv = gr[c4], sv = sr[c4];
            acc = fmaf(sv.x > 0.f ? gv.x : 0.f, ws[d][4 * c4], acc);
            acc = fmaf(sv.y > 0.f ? gv.y : 0.f, ws[d][4 * c4 + 1], acc);
            acc = fmaf(sv.z > 0.f ? gv.z : 0.f, ws[d][4 * c4 + 2], acc);
            acc = fmaf(sv.w > 0.f ? gv.w : 0.f, ws[d][4 * c4 + 3], acc);
        }
    }
    dx[i] = acc * out_scale;
}

// MaxPool1d(S, S) (S = sample_scale, 80 in every shipped checkpoint) over time of c[N][L][C] -> pooled[N][P][C] with the arg-max offset (first maximum)
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ c, float* __restrict__ pooled,
                                                       unsigned char* __restrict__ arg, int N, int L, int P, int C, int S) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;      // (n, w, ch)
    if (i >= (long long)N * P * C) return;
    const int ch = (int)(i % C);
    const long long nw = i / C;
    const int w = (int)(nw % P);
    const long long n = nw / P;
    const float* src = c + (n * L + (long long)w * S) * C + ch;
    float best = src[0];
    int bi = 0;
    for (int k = 1; k < S; ++k) {
        const float v = src[(long long)k * C];
        if (v > best) { best = v; bi = k; }
    }
    pooled[i] = best;
    arg[i] = (unsigned char)bi;
}

// gc[N][L][C] = 0 except gc[n][80w + arg][ch] = gpool[n][w][ch] * lrelu'(c at that position)
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ gpool, const unsigned char* __restrict__ arg,
                                                       const float* __restrict__ c, const float* __restrict__ pooled, float* __restrict__ gc,
                                                       int N, int L, int P, int C, int S) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)N * P * C) return;
    const int ch = (int)(i % C);
    const long long nw = i / C;
    const int w = (int)(nw % P);
    const long long n = nw / P;
    const long long pos = (n * L + (long long)w * S + arg[i]) * C + ch;
    const float s = pooled != nullptr ? pooled[i] : c[pos];       // the activation at the arg-max IS the pooled value
    gc[pos] = s > 0.f ? gpool[i] : 0.01f * gpool[i];
}

// x0[n][t][ch] = a[n][t][ch] + e[n][w(t)][ch]   (nearest upsample x80, shifted by rem_half, zero outside)
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ a, const float* __restrict__ e,
                                                           float* __restrict__ out, int N, int L, int P, int rem_half, int S) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;      // float4 index
    if (i >= (long long)N * L * 16) return;
    const int q = (int)(i & 15);
    const long long nt = i >> 4;
    const int t = (int)(nt % L);
    const long long n = nt / L;
    float4 v = ld4(a + nt * 64 + 4 * q);
    const int pos = t - rem_half;
    if (pos >= 0 && pos < S * P) {
        const float4 s = ld4(e + (n * P + pos / S) * 64 + 4 * q);
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
    }
    *reinterpret_cast<float4*>(out + nt * 64 + 4 * q) = v;
}

// the same for rows of C channels (any C): the standalone SemiGlobalBlock of models/stofnet.py:80 takes any width
__global__ __launch_bounds__(256) void upsample_add_c_kernel(const float* __restrict__ a, const float* __restrict__ e,
                                                             float* __restrict__ out, long long total, int L, int P, int rem_half, int S, int C) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % C);
    const long long nt = i / C;
    const int pos = (int)(nt % L) - rem_half;
    float v = a[i];
    if (pos >= 0 && pos < S * P) v += e[((nt / L) * P + pos / S) * C + ch];
    out[i] = v;
}

// ge[n][w][ch] = lrelu'(e) * sum_{t in window w} g[n][t][ch]
// One wave per (n, w): lane = (row subset part = lane >> 4, channel quad c4 = lane & 15); a lane adds the rows k = part mod 4
// of the window with 16-byte loads, the four subsets are combined in a fixed order.  (One thread per channel walking the 80
// rows with dword loads: 48 us for the 131 MB at batch 256.)
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ g, const float* __restrict__ e,
                                                           float* __restrict__ ge, int N, int L, int P, int rem_half, int S) {
    const long long nw = blockIdx.x * 4ll + (threadIdx.x >> 6);  // (n, w)
    if (nw >= (long long)N * P) return;
    const int lane = threadIdx.x & 63, part = lane >> 4, c4 = lane & 15;
    const int w = (int)(nw % P);
    const long long n = nw / P;
    const float* src = g + (n * L + rem_half + (long long)w * S) * 64 + 4 * c4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 5
    for (int k = part; k < S; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long long)k * 64);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x += __shfl_xor(s.x, 16); s.y += __shfl_xor(s.y, 16); s.z += __shfl_xor(s.z, 16); s.w += __shfl_xor(s.w, 16);
    s.x += __shfl_xor(s.x, 32); s.y += __shfl_xor(s.y, 32); s.z += __shfl_xor(s.z, 32); s.w += __shfl_xor(s.w, 32);
    if (part == 0) {
        const float4 ev = *reinterpret_cast<const float4*>(e + nw * 64 + 4 * c4);
        *reinterpret_cast<float4*>(ge + nw * 64 + 4 * c4) = make_float4(ev.x > 0.f ? s.x : 0.01f * s.x, ev.y > 0.f ? s.y : 0.01f * s.y,
                                                                        ev.z > 0.f ? s.z : 0.01f * s.z, ev.w > 0.f ? s.w : 0.01f * s.w);
    }
}

// ---- loss (main.py:228-232): target = 20 * blur7(onehot(gt)) / max(blur); MSE + lambda * mean|pred|
// One work-group per (row, segment of LOSS_SEG samples): with one group per row the launch had 256 groups of four waves,
// each walking 20,000 samples through LDS -- 39 us for 20 MB of output.
constexpr int LOSS_SEG = 2048;
__global__ __launch_bounds__(256) void loss_target_kernel(const long long* __restrict__ gt, int G, const float* __restrict__ taps,
                                                          float* __restrict__ target, int N, int M, float* __restrict__ tmax) {
    // scatter ones (index 0 cleared, negatives clamped: coords2mask), 7-tap blur with zero padding
    __shared__ float mask[LOSS_SEG + 6];                         // samples i0 - 3 .. i0 + LOSS_SEG + 2
    const int tid = threadIdx.x;
    const int segs = (M + LOSS_SEG - 1) / LOSS_SEG;
    const long long row = blockIdx.x / segs;
    const int i0 = (int)(blockIdx.x - row * segs) * LOSS_SEG;
    for (int i = tid; i < LOSS_SEG + 6; i += 256) mask[i] = 0.f;
    __syncthreads();
    for (int k = tid; k < G; k += 256) {
        long long idx = gt[row * G + k];
        if (idx < 0) idx = 0;
        const long long j = idx - i0 + 3;
        if (idx != 0 && idx < M && j >= 0 && j < LOSS_SEG + 6) mask[j] = 1.f;      // (index 0 is cleared)
    }
    __syncthreads();
    float mx = 0.f;
    for (int k = tid; k < LOSS_SEG; k += 256) {
        const int i = i0 + k;
        if (i >= M) break;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 7; ++d) { const int u = i + d - 3; if (u >= 0 && u < M) s = fmaf(taps[d], mask[k + d], s); }
        target[row * M + i] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    // mx >= 0: int order = float order; the word only grows, so a relaxed read first spares most groups the atomic
    if ((tid & 63) == 0 && __float_as_int(mx) > __hip_atomic_load(reinterpret_cast<int*>(tmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(reinterpret_cast<int*>(tmax), __float_as_int(mx));
}

// loss[0] += sum (pred - s*target)^2 / NM + lambda * sum |pred| / NM ; dpred = 2 (pred - s*target)/NM + lambda sign(pred)/NM
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ pred, float* __restrict__ target,
                                                        const float* __restrict__ tmax, float amplitude, float lambda,
                                                        long long count, float grad_scale, float* __restrict__ dpred,
                                                        double* __restrict__ loss) {
    const float tm = tmax[0];
    const double inv = 1.0 / (double)count;
    double part = 0.0;
    auto one = [&](float pv, float tv_in, float& tv, float& dp) {
        tv = tv_in / tm * amplitude;                             // main.py:230-231: /= max, then *= amplitude
        const float diff = pv - tv;
        part += ((double)diff * diff + (double)lambda * fabsf(pv)) * inv;
        const float sg = pv > 0.f ? 1.f : (pv < 0.f ? -1.f : 0.f);
        dp = (float)((2.0 * diff + (double)lambda * sg) * inv) * grad_scale;   // power-of-two scale: exact
    };
    // four elements per thread and step (16-byte accesses) where the three arrays allow it
    const bool vec = ((reinterpret_cast<size_t>(pred) | reinterpret_cast<size_t>(target) | reinterpret_cast<size_t>(dpred)) & 15) == 0;
    const long long nvec = vec ? count / 4 : 0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nvec; i += gridDim.x * 256ll) {
        const float4 pv = reinterpret_cast<const float4*>(pred)[i], tin = reinterpret_cast<const float4*>(target)[i];
        float4 tv, dp;
        one(pv.x, tin.x, tv.x, dp.x); one(pv.y, tin.y, tv.y, dp.y); one(pv.z, tin.z, tv.z, dp.z); one(pv.w, tin.w, tv.w, dp.w);
        reinterpret_cast<float4*>(target)[i] = tv;
        reinterpret_cast<float4*>(dpred)[i] = dp;
    }
    for (long long i = 4 * nvec + blockIdx.x * 256ll + threadIdx.x; i < count; i += gridDim.x * 256ll) {
        float tv, dp;
        one(pred[i], target[i], tv, dp);
        target[i] = tv;
        dpred[i] = dp;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    __shared__ double wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}

// torch.optim.AdamW step (decoupled weight decay), one flat parameter vector
// GUARD (split-fp16 training range guard): word[0] == epoch says grad_guard_kernel found a non-finite gradient or loss in THIS
// step -- the gradients then count as zero (and are zeroed), as if the step's backward had produced none, and the sticky flag
// word[1] is raised for the next host read (StofNetTrainer.raise_if_overflow)
template <bool GUARD>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, int* word, int epoch) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    float pv = p[i];
    float gv = g[i];
    if (GUARD) {
        if (*reinterpret_cast<volatile int*>(word) == epoch) {
            gv = 0.f;
            g[i] = 0.f;
            if (i == 0) atomicOr(word + 1, 1);
        }
    }
    pv *= 1.f - lr * wd;
    const float mv = beta1 * m[i] + (1.f - beta1) * gv;
    const float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - (lr / bc1) * (mv / denom);
}

// any non-finite gradient (exponent all ones) or loss in this step -> word[0] = epoch (epochs only grow: nothing to clear)
__global__ __launch_bounds__(256) void grad_guard_kernel(const float* __restrict__ g, long long n, const double* __restrict__ loss,
                                                         int* word, int epoch) {
    bool bad = false;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
    if (loss && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(*loss);
        bad |= (u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMax(word, epoch);
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long long n) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;        // four elements per thread where the arrays allow 16-byte accesses
    const bool vec = ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(out)) & 15) == 0;
    const long long n4 = vec ? n / 4 : 0;
    if (i < n4) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
    const long long j = 4 * n4 + i;                              // the rest (everything, for unaligned arrays)
    if (!vec) { for (long long k = j; k < n; k += (long long)gridDim.x * 256) out[k] = a[k] + b[k]; }
    else if (j < n) out[j] = a[j] + b[j];
}

// thread = (row, piece p of 8): channels 8 p .. 8 p + 7 = hi piece p + lo piece p of the split row, plus b's two float4
template <bool BSPLIT>                     // BSPLIT: b is a split-row tensor too
__global__ __launch_bounds__(256) void add_split_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ out, long long rows) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= rows * 8) return;
    const long long r = i >> 3;
    const int pc = (int)(i & 7);
    const float4 hi = ld4(a + r * 64 + 4 * pc), lo = ld4(a + r * 64 + 32 + 4 * pc);
    const unsigned hw[4] = {__float_as_uint(hi.x), __float_as_uint(hi.y), __float_as_uint(hi.z), __float_as_uint(hi.w)};
    const unsigned lw[4] = {__float_as_uint(lo.x), __float_as_uint(lo.y), __float_as_uint(lo.z), __float_as_uint(lo.w)};
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const half2v h = bits_h2w(hw[e]), l = bits_h2w(lw[e]);
        v[2 * e] = (float)h[0] + (float)l[0];
        v[2 * e + 1] = (float)h[1] + (float)l[1];
    }
    float4 b0, b1;
    if constexpr (BSPLIT) {
        const float4 bh = ld4(b + r * 64 + 4 * pc), bl = ld4(b + r * 64 + 32 + 4 * pc);
        const unsigned bhw[4] = {__float_as_uint(bh.x), __float_as_uint(bh.y), __float_as_uint(bh.z), __float_as_uint(bh.w)};
        const unsigned blw[4] = {__float_as_uint(bl.x), __float_as_uint(bl.y), __float_as_uint(bl.z), __float_as_uint(bl.w)};
        float w[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const half2v h = bits_h2w(bhw[e]), l = bits_h2w(blw[e]);
            w[2 * e] = (float)h[0] + (float)l[0];
            w[2 * e + 1] = (float)h[1] + (float)l[1];
        }
        b0 = make_float4(w[0], w[1], w[2], w[3]);
        b1 = make_float4(w[4], w[5], w[6], w[7]);
    } else {
        b0 = ld4(b + r * 64 + 8 * pc);
        b1 = ld4(b + r * 64 + 8 * pc + 4);
    }
    float* const o = out + r * 64 + 8 * pc;
    *reinterpret_cast<float4*>(o) = make_float4(v[0] + b0.x, v[1] + b0.y, v[2] + b0.z, v[3] + b0.w);
    *reinterpret_cast<float4*>(o + 4) = make_float4(v[4] + b1.x, v[5] + b1.y, v[6] + b1.z, v[7] + b1.w);
}

// fp32 rows [rows][64] -> split rows (thread = row, 4 channels)
__global__ __launch_bounds__(256) void to_split_rows_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= rows * 16) return;
    const long long r = i >> 4;
    const int qq = (int)(i & 15);
    uint2 hi, lo;
    split4(ld4(in + r * 64 + 4 * qq), hi, lo);
    char* const o = reinterpret_cast<char*>(out + r * 64);
    *reinterpret_cast<uint2*>(o + 8 * qq) = hi;
    *reinterpret_cast<uint2*>(o + 128 + 8 * qq) = lo;
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

namespace stof {
// Shared by the training entry point and by the inference forward (SemiGlobalBlock expand conv, stream mode).
int launch_conv_cl(const float* x, const float* w, const float* bias, const float* residual, const float* saved, float* y,
                   int64_t N, int64_t L, int32_t cin, int32_t cout, int32_t K, int32_t act, int32_t precision,
                   int32_t period, int32_t valid_len, hipStream_t stream, const int* run_if) {
    ConvParams p;
    p.run_if = run_if;
    p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.saved = saved; p.y = y;
    p.N = (int)N; p.L = (int)L; p.cin = cin; p.cout = cout; p.K = K; p.act = act;
    p.period = period; p.valid_len = valid_len;
    p.tiles_per_wf = (int)((L + CT - 1) / CT);
    const int64_t tiles = N * p.tiles_per_wf;
    if (tiles > 0x7fffffffLL || N * L > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    p.total_tiles = (int)tiles;
    static const bool fast16 = stof::body16_enabled();
    if (period == 0 && run_if == nullptr && fast16 && conv_cl16_ok(cin, cout, K, precision)) {
        p.tiles_per_wf = (int)((L + CT16 - 1) / CT16);
        const int64_t tiles16 = N * p.tiles_per_wf;
        if (tiles16 > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
        p.total_tiles = (int)tiles16;
        const int ob = cout / 64;
        int64_t g16 = 512 / ob;                      // persistent: 2 work-groups per CU
        if (g16 < 1) g16 = 1;
        if (g16 > tiles16) g16 = tiles16;
        const dim3 grid16((unsigned)g16, (unsigned)ob);
        if (K == 3) hipLaunchKernelGGL(conv_cl16_kernel<3>, grid16, dim3(256), 0, stream, p);
        else if (K == 5) hipLaunchKernelGGL(conv_cl16_kernel<5>, grid16, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(conv_cl16_kernel<7>, grid16, dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
    }
    const int oblocks = (cout + 63) / 64;
    int64_t gx = 512 / oblocks;                      // persistent: 2 work-groups per CU (71 KB of LDS each)
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
    const dim3 grid((unsigned)gx, (unsigned)oblocks);
    if (period > 0) {
        if (precision == STOF_PREC_F16X3)
            hipLaunchKernelGGL((conv_cl_kernel<STOF_PREC_F16X3, true>), grid, dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL((conv_cl_kernel<STOF_PREC_FP32, true>), grid, dim3(256), 0, stream, p);
    } else if (precision == STOF_PREC_F16X3) {
        hipLaunchKernelGGL((conv_cl_kernel<STOF_PREC_F16X3, false>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((conv_cl_kernel<STOF_PREC_FP32, false>), grid, dim3(256), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
}  // namespace stof

extern "C" int stof_train_conv(const float* x, const float* w_tapmajor, const float* bias, const float* residual,
                               const float* saved, float* y, int64_t N, int64_t L, int32_t cin, int32_t cout,
                               int32_t K, int32_t act, int32_t precision, void* stream) {
    if (N < 0 || L < 0 || cin < 1 || cout < 1 || K < 1 || K > 9 || !(K & 1)) return STOF_ERR_BAD_ARG;
    if (precision != STOF_PREC_FP32 && precision != STOF_PREC_F16X3) return STOF_ERR_BAD_ARG;
    if (N == 0 || L == 0) return STOF_OK;
    if (!x || !w_tapmajor || !y) return STOF_ERR_BAD_ARG;
    return stof::launch_conv_cl(x, w_tapmajor, bias, residual, saved, y, N, L, cin, cout, K, act, precision, 0, 0,
                                static_cast<hipStream_t>(stream), nullptr);
}

extern "C" size_t stof_train_repack_floats(int32_t cout, int32_t cin, int32_t K, int32_t transpose_flip, int32_t precision) {
    if (cout < 1 || cin < 1 || K < 1) return 0;
    if (precision != STOF_PREC_F16X3) return (size_t)cout * cin * K;
    const size_t A = transpose_flip ? cin : cout, B = transpose_flip ? cout : cin;
    return (size_t)K * A * ((B + 63) / 64 * 64);          // hi + lo halves = one float per (padded) element
}

extern "C" int stof_train_repack(const float* w, float* out, int32_t cout, int32_t cin, int32_t K, int32_t transpose_flip,
                                 int32_t precision, void* stream) {
    if (!w || !out || cout < 1 || cin < 1 || K < 1) return STOF_ERR_BAD_ARG;
    static const bool fast16 = stof::body16_enabled();
    if (fast16 && conv_cl16_ok(cin, cout, K, precision)) {
        // fragment image of conv_cl16_kernel (the conv entry point picks that kernel by the same test); same size
        const long long halves = (long long)K * cout * cin * 2;
        hipLaunchKernelGGL(repack_frag16_kernel, dim3(blocks_for(halves)), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                           reinterpret_cast<_Float16*>(out), cout, cin, K, transpose_flip);
        return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
    }
    if (precision == STOF_PREC_F16X3) {
        const long long total = (long long)stof_train_repack_floats(cout, cin, K, transpose_flip, precision);
        hipLaunchKernelGGL(repack_weights_split_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           w, reinterpret_cast<_Float16*>(out), cout, cin, K, transpose_flip);
    } else if (precision == STOF_PREC_FP32) {
        hipLaunchKernelGGL(repack_weights_kernel, dim3(blocks_for((long long)cout * cin * K)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), w, out, cout, cin, K, transpose_flip);
    } else {
        return STOF_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// persistent work-groups per (64 x 64) weight block: 2 per CU (fp32: 70 KB of LDS each; f16x3: 198 VGPRs -> 2 waves per SIMD)
static int wgrad_groups(int32_t cin, int32_t cout, int32_t precision) {
    (void)precision;
    const int blocks = ((cout + 63) / 64) * ((cin + 63) / 64);
    const int g = 512 / blocks;
    return g < 1 ? 1 : g;
}

extern "C" size_t stof_train_wgrad_workspace_bytes(int32_t cin, int32_t cout, int32_t K) {
    if (cin < 1 || cout < 1 || K < 1) return 0;
    const size_t cin_pad = (size_t)((cin + 63) / 64) * 64, cout_pad = (size_t)((cout + 63) / 64) * 64;
    return (size_t)wgrad_groups(cin, cout, STOF_PREC_F16X3) * ((size_t)K * cout_pad * cin_pad + cout_pad) * sizeof(float);
}

extern "C" int stof_train_wgrad(const float* x, const float* dy, float* dw, float* db, int64_t N, int64_t L,
                                int32_t cin, int32_t cout, int32_t K, float out_scale, int32_t precision,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 0 || L < 0 || cin < 1 || cout < 1 || K < 1 || K > 7 || !(K & 1)) return STOF_ERR_BAD_ARG;
    if (!dw) return STOF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0 || L == 0) {
        if (hipMemsetAsync(dw, 0, (size_t)cout * cin * K * sizeof(float), s) != hipSuccess) return STOF_ERR_HIP;
        if (db && hipMemsetAsync(db, 0, (size_t)cout * sizeof(float), s) != hipSuccess) return STOF_ERR_HIP;
        return STOF_OK;
    }
    if (!x || !dy || !workspace) return STOF_ERR_BAD_ARG;
    if (workspace_bytes < stof_train_wgrad_workspace_bytes(cin, cout, K)) return STOF_ERR_WORKSPACE;
    WgradParams p;
    p.x = x; p.dy = dy; p.N = (int)N; p.L = (int)L; p.cin = cin; p.cout = cout; p.K = K;
    p.cin_pad = (cin + 63) / 64 * 64; p.cout_pad = (cout + 63) / 64 * 64;
    if (precision != STOF_PREC_FP32 && precision != STOF_PREC_F16X3) return STOF_ERR_BAD_ARG;
    const int rows = precision == STOF_PREC_F16X3 ? WG_ROWS_H : WG_ROWS;
    p.tiles_per_wf = (int)((L + rows - 1) / rows);
    const int64_t tiles = N * p.tiles_per_wf;
    if (tiles > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    p.total_tiles = (int)tiles;
    int G = wgrad_groups(cin, cout, precision);
    if (G > tiles) G = (int)tiles;
    p.part = static_cast<float*>(workspace);
    p.dbpart = p.part + (size_t)G * K * p.cout_pad * p.cin_pad;
    const dim3 grid((unsigned)G, (unsigned)(p.cout_pad / 64), (unsigned)(p.cin_pad / 64));
    if (precision == STOF_PREC_F16X3) {
        if (K <= 3) hipLaunchKernelGGL(conv_wgrad_f16x3_kernel<3>, grid, dim3(256), 0, s, p);
        else if (K <= 5) hipLaunchKernelGGL(conv_wgrad_f16x3_kernel<5>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_wgrad_f16x3_kernel<7>, grid, dim3(256), 0, s, p);
    } else {
        if (K <= 3) hipLaunchKernelGGL(conv_wgrad_cl_kernel<3>, grid, dim3(256), 0, s, p);
        else if (K <= 5) hipLaunchKernelGGL(conv_wgrad_cl_kernel<5>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_wgrad_cl_kernel<7>, grid, dim3(256), 0, s, p);
    }
    const int total = K * p.cout_pad * p.cin_pad + p.cout_pad;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(64 * WRED_SLICES), 0, s, p.part, p.dbpart, dw, db, G, K, cout, cin,
                       p.cout_pad, p.cin_pad, out_scale);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// groups per layer of the batched launch: ~2 work-groups per CU over the whole launch, at least 8 per layer
static int wgrad_batch_groups(int count) {
    const int g = 2 * stof::device_cu_count() / (count < 1 ? 1 : count);
    return g < 8 ? 8 : g;
}

extern "C" size_t stof_train_wgrad_batch_workspace_bytes(int32_t count, int32_t K) {
    if (count < 1 || count > WGRAD_BATCH_MAX || K < 1) return 0;
    return (size_t)count * wgrad_batch_groups(count) * ((size_t)K * 64 * 64 + 64) * sizeof(float);
}

static int wgrad_batch_impl(const float* const* x, const float* const* dy, float* const* dw, float* const* db, int32_t count,
                            int64_t N, int64_t L, int32_t K, float out_scale, void* workspace, size_t workspace_bytes,
                            void* stream, uint32_t x_split, uint32_t dy_split) {
    if (!x || !dy || !dw || !db || count < 1 || count > WGRAD_BATCH_MAX || N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    if (K != 7 && K != 5 && K != 3) return STOF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WgradBatch b;
    for (int i = 0; i < count; ++i) {
        if (!dw[i] || !db[i] || ((N > 0 && L > 0) && (!x[i] || !dy[i]))) return STOF_ERR_BAD_ARG;
        b.x[i] = x[i]; b.dy[i] = dy[i]; b.dw[i] = dw[i]; b.db[i] = db[i];
    }
    if (N == 0 || L == 0) {
        for (int i = 0; i < count; ++i)
            if (hipMemsetAsync(dw[i], 0, (size_t)64 * 64 * K * sizeof(float), s) != hipSuccess ||
                hipMemsetAsync(db[i], 0, 64 * sizeof(float), s) != hipSuccess) return STOF_ERR_HIP;
        return STOF_OK;
    }
    if (!workspace || workspace_bytes < stof_train_wgrad_batch_workspace_bytes(count, K)) return STOF_ERR_WORKSPACE;
    b.N = (int)N; b.L = (int)L; b.K = K; b.count = count; b.out_scale = out_scale;
    b.x_split = x_split; b.dy_split = dy_split;
    b.tiles_per_wf = (int)((L + WG_ROWS_H - 1) / WG_ROWS_H);
    const int64_t tiles = N * b.tiles_per_wf;
    if (tiles > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    b.total_tiles = (int)tiles;
    int G = wgrad_batch_groups(count);
    if (G > tiles) G = (int)tiles;
    b.part = static_cast<float*>(workspace);
    const unsigned all = count >= 32 ? 0xffffffffu : ((1u << count) - 1u);
    // STOF_TRAIN_WGRAD_ASYNC (read per call: tests switch it): unset / 3 = the four-wave global_load_lds kernel on 16x16x32 MFMAs
    // (813 us on the box where the others were timed), 1 = the same on 32x32x16 (870), 2 = its eight-wave form (930), 0 = the
    // register-staged kernel (884)
    const char* const aenv = getenv("STOF_TRAIN_WGRAD_ASYNC");
    const int form = aenv ? atoi(aenv) : 3;
    if (form != 0 && K == 7 && (x_split & all) == all && (dy_split & all) == all && (form != 2 || (G >= 2 && (G & 1) == 0))) {
        // every operand is split rows: the global_load_lds kernel, G / 2 work-groups of eight waves per layer, G partials as before
        b.dbpart = b.part + (size_t)count * G * K * 64 * 64;
        static stof::LdsLimitOnce once[2];
        if (form == 2) {
            constexpr int lds_bytes = 3 * WA_BUF_BYTES;
            if (int st = once[0].ensure(reinterpret_cast<const void*>(&conv_wgrad_split_async_kernel<7, 2>), lds_bytes)) return st;
            hipLaunchKernelGGL((conv_wgrad_split_async_kernel<7, 2>), dim3((unsigned)(G / 2), (unsigned)count, 1), dim3(512), lds_bytes, s, b);
        } else if (form == 3) {                                          // the four-wave form on 16x16x32 MFMAs
            constexpr int lds_bytes = 2 * WA_BUF_BYTES;
            static stof::LdsLimitOnce once16;
            if (int st = once16.ensure(reinterpret_cast<const void*>(&conv_wgrad_split_async16_kernel<7>), lds_bytes)) return st;
            hipLaunchKernelGGL((conv_wgrad_split_async16_kernel<7>), dim3((unsigned)G, (unsigned)count, 1), dim3(256), lds_bytes, s, b);
        } else {
            constexpr int lds_bytes = 2 * WA_BUF_BYTES;
            if (int st = once[1].ensure(reinterpret_cast<const void*>(&conv_wgrad_split_async_kernel<7, 1>), lds_bytes)) return st;
            hipLaunchKernelGGL((conv_wgrad_split_async_kernel<7, 1>), dim3((unsigned)G, (unsigned)count, 1), dim3(256), lds_bytes, s, b);
        }
        const int total7 = K * 64 * 64 + 64;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((total7 + 63) / 64, (unsigned)count), dim3(64 * WRED_SLICES), 0, s, b, G);
        return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
    }
    b.dbpart = b.part + (size_t)count * G * K * 64 * 64;
    const dim3 grid((unsigned)G, (unsigned)count, 1);
    if (K == 3) hipLaunchKernelGGL(conv_wgrad_f16x3_batch_kernel<3>, grid, dim3(256), 0, s, b);
    else if (K == 5) hipLaunchKernelGGL(conv_wgrad_f16x3_batch_kernel<5>, grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL(conv_wgrad_f16x3_batch_kernel<7>, grid, dim3(256), 0, s, b);
    const int total = K * 64 * 64 + 64;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((total + 63) / 64, (unsigned)count), dim3(64 * WRED_SLICES), 0, s, b, G);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_wgrad_batch(const float* const* x, const float* const* dy, float* const* dw, float* const* db, int32_t count,
                                      int64_t N, int64_t L, int32_t K, float out_scale, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    return wgrad_batch_impl(x, dy, dw, db, count, N, L, K, out_scale, workspace, workspace_bytes, stream, 0u, 0u);
}
extern "C" int stof_train_wgrad_batch_split(const float* const* x, const float* const* dy, float* const* dw, float* const* db,
                                            int32_t count, uint32_t x_split, uint32_t dy_split, int64_t N, int64_t L, int32_t K,
                                            float out_scale, void* workspace, size_t workspace_bytes, void* stream) {
    for (int i = 0; i < count && i < 32; ++i)
        if ((((x_split >> i) & 1u) && x && (reinterpret_cast<size_t>(x[i]) & 15)) ||
            (((dy_split >> i) & 1u) && dy && (reinterpret_cast<size_t>(dy[i]) & 15))) return STOF_ERR_BAD_ARG;      // 16-byte pieces
    return wgrad_batch_impl(x, dy, dw, db, count, N, L, K, out_scale, workspace, workspace_bytes, stream, x_split, dy_split);
}

// fp32 rows [rows][64] -> split rows [64 x fp16 hi | 64 x fp16 lo] (conv12's output gradient for stof_train_wgrad_batch_split)
extern "C" int stof_train_to_split_rows(const float* in, float* out, int64_t rows, void* stream) {
    if (rows < 0) return STOF_ERR_BAD_ARG;
    if (rows == 0) return STOF_OK;
    if (!in || !out || ((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out)) & 15)) return STOF_ERR_BAD_ARG;
    hipLaunchKernelGGL(to_split_rows_kernel, dim3(blocks_for(rows * 16)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out,
                       (long long)rows);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// out = (hi + lo of the split rows a) + b   (the backward sweep's dL/dx_0 is a split-row tensor; the long skip adds the fp32 g6)
static int add_split_impl(const float* a_split, const float* b, float* out, int64_t rows, void* stream, bool b_split) {
    if (rows < 0) return STOF_ERR_BAD_ARG;
    if (rows == 0) return STOF_OK;
    if (!a_split || !b || !out) return STOF_ERR_BAD_ARG;
    if ((reinterpret_cast<size_t>(a_split) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(out)) & 15) return STOF_ERR_BAD_ARG;
    if (b_split)
        hipLaunchKernelGGL(add_split_kernel<true>, dim3(blocks_for(rows * 8)), dim3(256), 0, static_cast<hipStream_t>(stream), a_split, b, out,
                           (long long)rows);
    else
        hipLaunchKernelGGL(add_split_kernel<false>, dim3(blocks_for(rows * 8)), dim3(256), 0, static_cast<hipStream_t>(stream), a_split, b, out,
                           (long long)rows);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
extern "C" int stof_train_add_split(const float* a_split, const float* b, float* out, int64_t rows, void* stream) {
    return add_split_impl(a_split, b, out, rows, stream, false);
}
// out = (a_split) + (b_split), both split-row tensors, out fp32
extern "C" int stof_train_add_split2(const float* a_split, const float* b_split, float* out, int64_t rows, void* stream) {
    return add_split_impl(a_split, b_split, out, rows, stream, true);
}

extern "C" int stof_train_conv1(const float* x, const float* w, const float* b, float* y, int64_t N, int64_t L, void* stream) {
    if (N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    if (N == 0 || L == 0) return STOF_OK;
    if (!x || !w || !b || !y) return STOF_ERR_BAD_ARG;
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3((unsigned)(N * ((L + C1_ROWS - 1) / C1_ROWS))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, w, b, y, (int)N, (int)L);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" size_t stof_train_conv1_wgrad_workspace_bytes(void) { return (size_t)C1_COPIES * 640 * sizeof(float); }

extern "C" int stof_train_conv1_wgrad(const float* x, const float* g, const float* saved, float* dw, float* db, int64_t N,
                                      int64_t L, float out_scale, void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    if (!dw || !db) return STOF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0 || L == 0) {
        if (hipMemsetAsync(dw, 0, 64 * 9 * sizeof(float), s) != hipSuccess || hipMemsetAsync(db, 0, 64 * sizeof(float), s) != hipSuccess)
            return STOF_ERR_HIP;
        return STOF_OK;
    }
    if (!x || !g || !saved || !workspace) return STOF_ERR_BAD_ARG;
    if (workspace_bytes < stof_train_conv1_wgrad_workspace_bytes()) return STOF_ERR_WORKSPACE;
    if (N * L > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    float* copies = static_cast<float*>(workspace);
    const int64_t chunks = (N * L + C1_CHUNK - 1) / C1_CHUNK;
    const int grid = (int)(chunks < C1_COPIES ? chunks : C1_COPIES);
    hipLaunchKernelGGL(conv1_wgrad_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, g, saved, copies, (int)N, (int)L);
    hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(10), dim3(1024), 0, s, copies, grid, dw, db, out_scale);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_conv1_dgrad(const float* g, const float* saved, const float* w, float* dx, int64_t N, int64_t L,
                                      float out_scale, void* stream) {
    if (N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    if (N == 0 || L == 0) return STOF_OK;
    if (!g || !saved || !w || !dx) return STOF_ERR_BAD_ARG;
    if (N * L > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(conv1_dgrad_kernel, dim3((unsigned)((N * L + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g, saved, w, dx, (int)N, (int)L, out_scale);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_pool(const float* c, float* pooled, uint8_t* arg, int64_t N, int64_t L, int64_t P, int32_t C,
                               int32_t scale, void* stream) {
    if (N < 0 || L < 0 || P < 0 || C < 1 || scale < 1 || scale > 256 || P * scale > L) return STOF_ERR_BAD_ARG;
    if (N * P == 0) return STOF_OK;
    if (!c || !pooled || !arg) return STOF_ERR_BAD_ARG;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(blocks_for(N * P * C)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       c, pooled, arg, (int)N, (int)L, (int)P, C, scale);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_pool_bwd(const float* gpool, const uint8_t* arg, const float* c, const float* pooled, float* gc,
                                   int64_t N, int64_t L, int64_t P, int32_t C, int32_t scale, void* stream) {
    if (N < 0 || L < 0 || P < 0 || C < 1 || scale < 1 || scale > 256 || P * scale > L) return STOF_ERR_BAD_ARG;
    if (N * L == 0) return STOF_OK;
    if (!gpool || !arg || (!c && !pooled) || !gc) return STOF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(gc, 0, (size_t)N * L * C * sizeof(float), s) != hipSuccess) return STOF_ERR_HIP;
    if (N * P > 0)
        hipLaunchKernelGGL(pool_bwd_kernel, dim3(blocks_for(N * P * C)), dim3(256), 0, s, gpool, arg, c, c ? nullptr : pooled, gc, (int)N, (int)L,
                           (int)P, C, scale);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// SemiGlobalBlock backward: contract_conv's weight / bias gradient straight from the pool's sparse gradient (see
// sgb_contract_wgrad_kernel).  STOF_ERR_UNSUPPORTED for shapes it does not take (C not a multiple of 128, cin != 64,
// S > 92): the caller then builds the dense gradient (stof_train_pool_bwd) and calls stof_train_wgrad.
extern "C" size_t stof_train_sgb_wgrad_workspace_bytes(int32_t C) {
    const int G = stof::device_cu_count() * 2 / (C / SGBW_CH > 0 ? C / SGBW_CH : 1);
    return (size_t)(G > 0 ? G : 1) * C * 321 * sizeof(float);
}

extern "C" int stof_train_sgb_contract_wgrad(const float* gpool, const uint8_t* arg, const float* pooled, const float* a1, float* dw,
                                             float* db, int64_t N, int64_t L, int64_t P, int32_t C, int32_t scale, float out_scale,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 0 || L < 0 || P < 0 || C < 1 || scale < 1 || P * scale > L) return STOF_ERR_BAD_ARG;
    if (C % SGBW_CH != 0 || scale + 4 > SGBW_ROWS || L > 0x7fffffff / 64) return STOF_ERR_UNSUPPORTED;
    if (!dw || !db || !workspace) return STOF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int G = stof::device_cu_count() * 2 / (C / SGBW_CH);
    if (G < 1) G = 1;
    if (workspace_bytes < (size_t)G * C * 321 * sizeof(float)) return STOF_ERR_WORKSPACE;
    if (N * P == 0) {
        if (hipMemsetAsync(dw, 0, (size_t)C * 320 * sizeof(float), s) != hipSuccess ||
            hipMemsetAsync(db, 0, (size_t)C * sizeof(float), s) != hipSuccess) return STOF_ERR_HIP;
        return STOF_OK;
    }
    if (!gpool || !arg || !pooled || !a1) return STOF_ERR_BAD_ARG;
    if (N * P < G) G = (int)(N * P);
    SgbWgradParams p;
    p.gpool = gpool; p.arg = arg; p.pooled = pooled; p.a1 = a1;
    p.part = static_cast<float*>(workspace);
    p.dbpart = p.part + (size_t)G * C * 320;
    p.nwin = N * P; p.L = (int)L; p.P = (int)P; p.C = C; p.S = scale; p.G = G;
    const size_t lds = (size_t)SGBW_NBUF * SGBW_BUF_F * sizeof(float);
    static stof::LdsLimitOnce once;
    if (int st = once.ensure(reinterpret_cast<const void*>(&sgb_contract_wgrad_kernel), 160 * 1024)) return st;
    hipLaunchKernelGGL(sgb_contract_wgrad_kernel, dim3(G, C / SGBW_CH), dim3(512), lds, s, p);
    hipLaunchKernelGGL(sgb_wgrad_reduce_kernel, dim3((unsigned)((C * 321 + 63) / 64)), dim3(64 * WRED_SLICES), 0, s, p.part, p.dbpart, dw, db, G, C,
                       out_scale);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// SemiGlobalBlock backward: dL/da1 = resid (may be NULL) + conv_transpose(gc, contract_conv.weight) from the pool's sparse
// gradient (see sgb_contract_dgrad_kernel); out[N, L, 64], fixed summation order.  workspace: C * 320 floats (the weights
// transposed).  STOF_ERR_UNSUPPORTED when C is not a multiple of 64, C > 512, scale > 88, scale < 4 or P == 0 (the caller then
// takes the dense route).
extern "C" size_t stof_train_sgb_dgrad_workspace_bytes(int32_t C) { return (size_t)(C > 0 ? C : 0) * 320 * sizeof(float); }

extern "C" int stof_train_sgb_contract_dgrad(const float* gpool, const uint8_t* arg, const float* pooled, const float* weight,
                                             const float* resid, float* out, int64_t N, int64_t L, int64_t P, int32_t C, int32_t scale,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 0 || L < 0 || P < 0 || C < 1 || scale < 1 || P * scale > L) return STOF_ERR_BAD_ARG;
    if (C % 64 != 0 || C > SGBD_MAXC || scale > SGBD_MAXS || scale < 4 || P == 0 || L > 0x7fffffff / 64 || N * P > 0x7fffffffLL)
        return STOF_ERR_UNSUPPORTED;
    if (N == 0) return STOF_OK;
    if (!out || !weight || !gpool || !arg || !pooled || !workspace) return STOF_ERR_BAD_ARG;
    if (workspace_bytes < stof_train_sgb_dgrad_workspace_bytes(C)) return STOF_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* const wt = static_cast<float*>(workspace);
    hipLaunchKernelGGL(sgb_wt_repack_kernel, dim3(blocks_for((long long)C * 320)), dim3(256), 0, s, weight, wt, C);
    SgbDgradParams p;
    p.gpool = gpool; p.arg = arg; p.pooled = pooled; p.wt = wt; p.resid = resid; p.out = out;
    p.nwin = N * P; p.L = (int)L; p.P = (int)P; p.C = C; p.S = scale;
    hipLaunchKernelGGL(sgb_contract_dgrad_kernel, dim3((unsigned)((N * P + SGBD_WAVES - 1) / SGBD_WAVES)), dim3(64 * SGBD_WAVES), 0, s, p);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// Data gradient of conv_last: out[N, L, 64] from dz[N, L, r] and conv_last.weight[r][64][3] (exact fp32).  r = 4 or 10;
// other factors return STOF_ERR_UNSUPPORTED (the caller uses stof_train_conv with the repacked weights).
static int conv_last_dgrad_impl(const float* dz, const float* weight, float* out, int64_t N, int64_t L, int32_t r, void* stream, bool split) {
    if (N < 0 || L < 0 || r < 1) return STOF_ERR_BAD_ARG;
    if (r != 4 && r != 10) return STOF_ERR_UNSUPPORTED;
    if (N == 0 || L == 0) return STOF_OK;
    if (!dz || !weight || !out) return STOF_ERR_BAD_ARG;
    const int64_t groups = N * ((L + CLD_ROWS - 1) / CLD_ROWS);
    if (groups > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (split) {
        if (r == 10) hipLaunchKernelGGL((conv_last_dgrad_kernel<10, true>), dim3((unsigned)groups), dim3(256), 0, s, dz, weight, out, (int)N, (int)L);
        else hipLaunchKernelGGL((conv_last_dgrad_kernel<4, true>), dim3((unsigned)groups), dim3(256), 0, s, dz, weight, out, (int)N, (int)L);
    } else {
        if (r == 10) hipLaunchKernelGGL(conv_last_dgrad_kernel<10>, dim3((unsigned)groups), dim3(256), 0, s, dz, weight, out, (int)N, (int)L);
        else hipLaunchKernelGGL(conv_last_dgrad_kernel<4>, dim3((unsigned)groups), dim3(256), 0, s, dz, weight, out, (int)N, (int)L);
    }
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
extern "C" int stof_train_conv_last_dgrad(const float* dz, const float* weight, float* out, int64_t N, int64_t L, int32_t r, void* stream) {
    return conv_last_dgrad_impl(dz, weight, out, N, L, r, stream, false);
}
// the same with out[N, L] written as split rows (stof_train_sweep_bwd_split and stof_train_wgrad_batch_split read it as such)
extern "C" int stof_train_conv_last_dgrad_split(const float* dz, const float* weight, float* out, int64_t N, int64_t L, int32_t r, void* stream) {
    return conv_last_dgrad_impl(dz, weight, out, N, L, r, stream, true);
}

extern "C" int stof_train_upsample_add(const float* a, const float* e, float* out, int64_t N, int64_t L, int64_t P,
                                       int32_t rem_half, int32_t scale, void* stream) {
    if (N < 0 || L < 0 || P < 0 || scale < 1 || rem_half < 0 || rem_half + P * scale > L) return STOF_ERR_BAD_ARG;
    if (N * L == 0) return STOF_OK;
    if (!a || !out || (!e && P > 0)) return STOF_ERR_BAD_ARG;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(blocks_for(N * L * 16)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       a, e, out, (int)N, (int)L, (int)P, rem_half, scale);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_upsample_add_c(const float* a, const float* e, float* out, int64_t N, int64_t L, int64_t P,
                                         int32_t rem_half, int32_t scale, int32_t C, void* stream) {
    if (N < 0 || L < 0 || P < 0 || C < 1 || scale < 1 || rem_half < 0 || rem_half + P * scale > L) return STOF_ERR_BAD_ARG;
    if (N * L == 0) return STOF_OK;
    if (!a || !out || (!e && P > 0)) return STOF_ERR_BAD_ARG;
    if (L > 0x7fffffffLL || P * scale > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    const long long total = (long long)N * L * C;
    hipLaunchKernelGGL(upsample_add_c_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       a, e, out, total, (int)L, (int)P, rem_half, scale, C);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_upsample_bwd(const float* g, const float* e, float* ge, int64_t N, int64_t L, int64_t P,
                                       int32_t rem_half, int32_t scale, void* stream) {
    if (N < 0 || L < 0 || P < 0 || scale < 1 || rem_half < 0 || rem_half + P * scale > L) return STOF_ERR_BAD_ARG;
    if (N * P == 0) return STOF_OK;
    if (!g || !e || !ge) return STOF_ERR_BAD_ARG;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((N * P + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g, e, ge, (int)N, (int)L, (int)P, rem_half, scale);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// The loss of main.py:228-232 in two halves, so that a batch sharded over ranks can MAX-all-reduce the maximum of the
// blurred target in between (`masks_true_blur /= masks_true_blur.max()` is a maximum over the WHOLE batch, main.py:230).
extern "C" int stof_train_loss_target(const int64_t* gt_idx, int64_t G, const float* taps7, int64_t N, int64_t M,
                                      float* target, float* tmax, void* stream) {
    if (N < 0 || M < 0 || G < 0) return STOF_ERR_BAD_ARG;
    if (N * M == 0) return STOF_OK;
    if (!gt_idx || !taps7 || !target || !tmax) return STOF_ERR_BAD_ARG;
    if (N * ((M + LOSS_SEG - 1) / LOSS_SEG) > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(tmax, 0, sizeof(float), s) != hipSuccess) return STOF_ERR_HIP;
    hipLaunchKernelGGL(loss_target_kernel, dim3((unsigned)(N * ((M + LOSS_SEG - 1) / LOSS_SEG))), dim3(256), 0, s,
                       reinterpret_cast<const long long*>(gt_idx), (int)G, taps7, target, (int)N, (int)M, tmax);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_loss_grad(const float* pred, float* target, const float* tmax, int64_t N, int64_t M, float amplitude,
                                    float lambda, float grad_scale, float* dpred, double* loss, void* stream) {
    if (N < 0 || M < 0) return STOF_ERR_BAD_ARG;
    if (N * M == 0) return STOF_OK;
    if (!pred || !target || !tmax || !dpred || !loss) return STOF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(loss, 0, sizeof(double), s) != hipSuccess) return STOF_ERR_HIP;
    // one double atomic per block on ONE address (~90 per microsecond): 512 grid-striding blocks, not thousands
    const unsigned lblocks = blocks_for(N * M) < 512u ? blocks_for(N * M) : 512u;
    hipLaunchKernelGGL(loss_grad_kernel, dim3(lblocks), dim3(256), 0, s, pred, target, tmax, amplitude, lambda,
                       (long long)(N * M), grad_scale, dpred, loss);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// single-process form: both halves back to back
extern "C" int stof_train_loss(const float* pred, const int64_t* gt_idx, int64_t G, const float* taps7, int64_t N, int64_t M,
                               float amplitude, float lambda, float grad_scale, float* target, float* tmax, float* dpred,
                               double* loss, void* stream) {
    if (!pred || !dpred || !loss) return (N * M == 0 && N >= 0 && M >= 0 && G >= 0) ? STOF_OK : STOF_ERR_BAD_ARG;
    const int st = stof_train_loss_target(gt_idx, G, taps7, N, M, target, tmax, stream);
    if (st != STOF_OK) return st;
    return stof_train_loss_grad(pred, target, tmax, N, M, amplitude, lambda, grad_scale, dpred, loss, stream);
}

extern "C" int stof_train_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
    if (n < 0 || step < 1) return STOF_ERR_BAD_ARG;
    if (n == 0) return STOF_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return STOF_ERR_BAD_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), params,
                       const_cast<float*>(grads), exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                       (float)sqrt(bc2), static_cast<int*>(nullptr), 0);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_adamw_guarded(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                        float beta1, float beta2, float eps, float weight_decay, int64_t step, const double* loss,
                                        int32_t* guard_words, void* stream) {
    if (n < 0 || step < 1 || step > 0x7fffffffLL) return STOF_ERR_BAD_ARG;
    if (n == 0) return STOF_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !guard_words) return STOF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const unsigned scan_blocks = blocks_for(n) < 1024u ? blocks_for(n) : 1024u;
    hipLaunchKernelGGL(grad_guard_kernel, dim3(scan_blocks), dim3(256), 0, s, grads, (long long)n, loss, guard_words, (int)step);
    hipLaunchKernelGGL(adamw_kernel<true>, dim3(blocks_for(n)), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (long long)n, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), guard_words, (int)step);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (n < 0) return STOF_ERR_BAD_ARG;
    if (n == 0) return STOF_OK;
    if (!a || !b || !out) return STOF_ERR_BAD_ARG;
    const bool vec = ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(out)) & 15) == 0;
    const long long threads = vec ? (n / 4 > 4 ? n / 4 : 4) : n;                 // (the kernel makes the same choice)
    hipLaunchKernelGGL(add_kernel, dim3(blocks_for(threads)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, out, (long long)n);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
