// StofNet.forward (models/stofnet.py:42-67) as three gfx950 kernels:
//
//   sgb_contract_pool : x -> relu(conv1) -> contract_conv 64->512 k5 (MFMA) -> lrelu
//                       -> max-pool 80  => pooled[N][P][512]        (models/stofnet.py:45,100-103)
//   sgb_expand        : pooled -> expand_conv 512->64 k5 -> lrelu   => sgb[N][P][64]   (:106-107)
//   body_sweep        : x, sgb -> relu(conv1) + upsample/pad(sgb) (:45,108-115) -> conv2..conv12
//                       with the residual pattern of :51-62 -> conv_last -> SampleShuffle1D store
//                       (:65, utils/sample_shuffle.py:10-28)        => y[N][L*r]
//
// All activations between conv1 and the shuffle store live in LDS: body_sweep walks a
// stream of waveforms left to right in steps of S rows; every layer keeps a frontier that
// lags 3 rows per conv behind the previous one, the residual stream lives in ring X
// (updated in place), intermediates in ring Y, and layer weights stream through a
// double-buffered LDS chunk.  No halo is recomputed and nothing but x, sgb and y touches
// HBM.  The schedule is emulated in numpy by oracle/sweep_emulator.py.
//
// Arithmetic: exact fp32 on v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, bit-for-bit
// fp32), the parity baseline mode (STOF_PREC_FP32).
#include <hip/hip_runtime.h>
#include "stof_common.h"

using namespace stof;

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int LAG_LAST = 34;      // frontier lag of conv_last: 11 convs x 3 + 1

__device__ __forceinline__ int layer_lag(int j) { return j <= 11 ? 3 * j : LAG_LAST; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ floatx16 mfma4(const float4 a, const float4 b, floatx16 c) {
    // four K=2 steps: lane (i, h) holds channels 4h..4h+3 of an 8-channel group, step s
    // contracts channels {s, 4+s}
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// ----------------------------------------------------------------------------------
// body sweep
// ----------------------------------------------------------------------------------
struct BodyParams {
    const float* x;        // [N][L]
    const float* sgb;      // [N][P][64] or nullptr
    float* y;              // [N][L*r]
    const float* c1;       // [64][10]
    const float* bias;     // [13][64]
    const float* chunks;   // [BODY_NCHUNK][BODY_CHUNK_F]
    int N, L, r, P, rem_half, wf_per_wg;
};

template <int S, int RING, int RAWRING>
struct BodyLds {
    static constexpr int X = 0;
    static constexpr int Y = X + RING * ROWF;
    static constexpr int W = Y + RING * ROWF;
    static constexpr int RAW = W + 2 * BODY_CHUNK_F;
    static constexpr int BIAS = RAW + RAWRING;
    static constexpr int TOTAL = BIAS + 13 * 64;
    static constexpr size_t BYTES = (size_t)TOTAL * sizeof(float);
};

template <int S, int RING, int RAWRING>
__global__ __launch_bounds__(256, 1) void body_sweep_kernel(const BodyParams p) {
    static_assert((RING & (RING - 1)) == 0 && (RAWRING & (RAWRING - 1)) == 0, "rings are powers of two");
    static_assert(S % 64 == 0 && S + 36 <= RING && S + 42 <= RAWRING, "ring must hold the live span");
    using Lds = BodyLds<S, RING, RAWRING>;
    constexpr int NT = S / 64;                   // N-tiles (32 rows) per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Xr = smem + Lds::X;
    float* const Yr = smem + Lds::Y;
    float* const wbuf = smem + Lds::W;
    float* const rawr = smem + Lds::RAW;
    float* const biasl = smem + Lds::BIAS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int mi = wave & 1, ni = wave >> 1;
    const int ln = lane & 31, lh = lane >> 5;

    const int n0 = blockIdx.x * p.wf_per_wg;
    const int n1 = min(p.N, n0 + p.wf_per_wg);
    if (n0 >= n1) return;
    const int L = p.L, r = p.r;
    const int Lp = L + GAP;
    const int gend = (n1 - n0) * Lp;             // local stream rows [0, gend)

    // ---- one-time setup: zero rings, biases to LDS, conv1 taps to registers
    for (int i = tid; i < Lds::W; i += 256) smem[i] = 0.f;
    for (int i = tid; i < RAWRING; i += 256) rawr[i] = 0.f;
    for (int i = tid; i < 13 * 64; i += 256) biasl[i] = p.bias[i];
    const int cq = tid & 15, rl = tid >> 4;
    float w1[4][9], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int d = 0; d < 9; ++d) w1[i][d] = p.c1[(4 * cq + i) * 10 + d];
        b1[i] = p.c1[(4 * cq + i) * 10 + 9];
    }

    // relu(conv1(x)) + SemiGlobalBlock contribution for stream rows [rstart, rstart+S) -> ring dst
    auto x0_pass = [&](float* dst, int rstart) {
#pragma unroll 2
        for (int it = 0; it < S / 16; ++it) {
            const int g = rstart + rl + 16 * it;
            const bool inrange = (g >= 0) && (g < gend);
            const unsigned nl = inrange ? (unsigned)g / (unsigned)Lp : 0u;
            const int t = g - (int)nl * Lp;
            const bool valid = inrange && (t < L);
            float xs[9];
#pragma unroll
            for (int d = 0; d < 9; ++d) xs[d] = rawr[(g + d - 4) & (RAWRING - 1)];
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = b1[i];
#pragma unroll
                for (int d = 0; d < 9; ++d) a = fmaf(w1[i][d], xs[d], a);
                v[i] = fmaxf(a, 0.f);
            }
            if (p.sgb != nullptr && valid) {
                const int pos = t - p.rem_half;
                if (pos >= 0 && pos < SGB_SCALE * p.P) {
                    const int w = pos / SGB_SCALE;
                    const float4 s = ld4(p.sgb + ((size_t)(n0 + nl) * p.P + w) * NF + 4 * cq);
                    v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
                }
            }
            float4 o = valid ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            st4(dst + (g & (RING - 1)) * ROWF + 4 * cq, o);
        }
    };

    // weight-chunk staging: global -> registers early, registers -> LDS late
    const float4* const gchunks = reinterpret_cast<const float4*>(p.chunks);
    constexpr int CHUNK_V4 = BODY_CHUNK_F / 4;   // 576
    float4 st0, st1, st2;
    auto stage_load = [&](int c) {
        const float4* src = gchunks + (size_t)c * CHUNK_V4;
        st0 = src[tid];
        st1 = src[tid + 256];
        if (tid < CHUNK_V4 - 512) st2 = src[tid + 512];
    };
    auto stage_store = [&](int buf) {
        float4* dstv = reinterpret_cast<float4*>(wbuf + buf * BODY_CHUNK_F);
        dstv[tid] = st0;
        dstv[tid + 256] = st1;
        if (tid < CHUNK_V4 - 512) dstv[tid + 512] = st2;
    };

    stage_load(0);
    stage_store(0);
    // the per-step raw load covers rows [F+4-S, F+4); rows 0..3 of the stream precede the first one
    if (tid < 4 && tid < L) rawr[tid] = p.x[(size_t)n0 * L + tid];
    __syncthreads();

    const int nsteps = (gend - GAP + LAG_LAST + S - 1) / S;
    for (int step = 1; step <= nsteps; ++step) {
        const int F = step * S;
        // raw waveform rows [F+4-S, F+4) into the raw ring (zero in gaps / outside the range)
        if (tid < S) {
            const int g = F + 4 - S + tid;
            float v = 0.f;
            if (g >= 0 && g < gend) {
                const unsigned nl = (unsigned)g / (unsigned)Lp;
                const int t = g - (int)nl * Lp;
                if (t < L) v = p.x[(size_t)(n0 + nl) * L + t];
            }
            rawr[g & (RAWRING - 1)] = v;
        }
        __syncthreads();
        x0_pass(Xr, F - S);                       // sweep layer 0
        __syncthreads();

        int c = 0;                                // chunk index within the step
        for (int j = 1; j <= 12; ++j) {
            if (j == 11) {                        // long skip: seed the destination with x0
                x0_pass(Yr, F - S - 33);
                __syncthreads();
            }
            const bool last = (j == 12);
            const bool reads_x = (j & 1) || (j == 11);
            const float* const src = reads_x ? Xr : Yr;
            const int K = last ? 3 : 7, half = K >> 1;
            const int R0 = F - S - layer_lag(j);
            const bool active = !(last && mi == 1 && r <= 32);
            floatx16 acc[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

            const int nchunk = 2 * K;
            for (int cc = 0; cc < nchunk; ++cc, ++c) {
                const int d = cc >> 1, hh = cc & 1;
                const int nextc = (c + 1 == BODY_NCHUNK) ? 0 : c + 1;
                stage_load(nextc);
                if (active) {
                    const float* wb = wbuf + (c & 1) * BODY_CHUNK_F + (32 * mi + ln) * WROWF + 4 * lh;
                    const float* brow[NT];
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const int g = R0 + 32 * (NT * ni + k) + ln + d - half;
                        brow[k] = src + (g & (RING - 1)) * ROWF + 32 * hh + 4 * lh;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = ld4(wb + 8 * q);
#pragma unroll
                        for (int k = 0; k < NT; ++k) {
                            const float4 b = ld4(brow[k] + 8 * q);
                            acc[k] = mfma4(a, b, acc[k]);
                        }
                    }
                }
                if (cc == nchunk - 1) {
                    // ---- epilogue of sweep layer j (the destination ring is not read by this layer)
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const int g = R0 + 32 * (NT * ni + k) + ln;
                        const bool inrange = (g >= 0) && (g < gend);
                        const unsigned nl = inrange ? (unsigned)g / (unsigned)Lp : 0u;
                        const int t = g - (int)nl * Lp;
                        const bool valid = inrange && (t < L);
                        if (!last) {
                            const bool to_y = (j & 1);          // odd sweep layers (conv2,4,..,10, conv12) write ring Y
                            float* const dst = to_y ? Yr : Xr;
                            const bool inplace = !(j & 1) || (j == 11);
                            const bool act = (j & 1) && (j != 11);
                            float* const drow = dst + (g & (RING - 1)) * ROWF + 32 * mi + 4 * lh;
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) {
                                const float4 bb = ld4(biasl + j * 64 + 32 * mi + 8 * gg + 4 * lh);
                                float4 v = make_float4(acc[k][4 * gg] + bb.x, acc[k][4 * gg + 1] + bb.y,
                                                       acc[k][4 * gg + 2] + bb.z, acc[k][4 * gg + 3] + bb.w);
                                if (act) {
                                    v.x = v.x > 0.f ? v.x : 0.01f * v.x;
                                    v.y = v.y > 0.f ? v.y : 0.01f * v.y;
                                    v.z = v.z > 0.f ? v.z : 0.01f * v.z;
                                    v.w = v.w > 0.f ? v.w : 0.01f * v.w;
                                }
                                if (inplace) {
                                    const float4 o = ld4(drow + 8 * gg);
                                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                                }
                                if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
                                st4(drow + 8 * gg, v);
                            }
                        } else if (active && valid) {
                            // conv_last + SampleShuffle1D: out[n][t*r + k] = conv_last[n][k][t]
                            float* const orow = p.y + ((size_t)(n0 + nl) * L + t) * r;
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) {
                                const int c0 = 32 * mi + 8 * gg + 4 * lh;
                                const float4 bb = ld4(biasl + 12 * 64 + c0);
                                const float vv[4] = {acc[k][4 * gg] + bb.x, acc[k][4 * gg + 1] + bb.y,
                                                     acc[k][4 * gg + 2] + bb.z, acc[k][4 * gg + 3] + bb.w};
                                if ((r & 3) == 0 && c0 + 3 < r) {
                                    st4(orow + c0, make_float4(vv[0], vv[1], vv[2], vv[3]));
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (c0 + e < r) orow[c0 + e] = vv[e];
                                }
                            }
                        }
                    }
                }
                stage_store((c + 1) & 1);
                __syncthreads();
            }
        }
    }
}

// ----------------------------------------------------------------------------------
// SemiGlobalBlock contracting path: relu(conv1) -> conv 64->512 k5 -> lrelu -> maxpool 80
// One work-group = NW pooling windows of one waveform.  Time sits on the MFMA M axis so
// the pool is an in-lane max over accumulator registers plus one cross-half shuffle.
// ----------------------------------------------------------------------------------
struct SgbParams {
    const float* x;        // [N][L]
    float* pooled;         // [N][P][512]
    const float* c1;       // [64][10]
    const float* cbias;    // [512]
    const float* chunks;   // [SGB_NCHUNK][SGB_CHUNK_F]
    int N, L, P, tiles_per_wf;
};

template <int NW>
__global__ __launch_bounds__(256, 1) void sgb_contract_pool_kernel(const SgbParams p) {
    static_assert(NW % 2 == 0, "80*NW must be a multiple of 32");
    constexpr int ROWS = SGB_SCALE * NW;          // output rows of the tile
    constexpr int MT = ROWS / 32;
    constexpr int TR = ROWS + 4;                  // conv1 rows needed (k5: +-2)
    constexpr int RAWN = TR + 8;                  // raw samples needed (k9: +-4)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const act = smem;                      // [TR][ROWF]
    float* const wbuf = act + TR * ROWF;          // [2][SGB_CHUNK_F]
    float* const raw = wbuf + 2 * SGB_CHUNK_F;    // [RAWN]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;
    const int n = blockIdx.x / p.tiles_per_wf;
    const int w0 = (blockIdx.x - n * p.tiles_per_wf) * NW;
    const int L = p.L;
    const int tbase = SGB_SCALE * w0 - 2;         // time of act row 0

    for (int i = tid; i < RAWN; i += 256) {
        const int t = tbase - 4 + i;
        raw[i] = (t >= 0 && t < L) ? p.x[(size_t)n * L + t] : 0.f;
    }
    const float4* const gchunks = reinterpret_cast<const float4*>(p.chunks);
    constexpr int CHUNK_V4 = SGB_CHUNK_F / 4;     // 1152
    float4 stg[5];
    auto stage_load = [&](int c) {
        const float4* src = gchunks + (size_t)c * CHUNK_V4;
#pragma unroll
        for (int i = 0; i < 4; ++i) stg[i] = src[tid + 256 * i];
        if (tid < CHUNK_V4 - 1024) stg[4] = src[tid + 1024];
    };
    auto stage_store = [&](int buf) {
        float4* dstv = reinterpret_cast<float4*>(wbuf + buf * SGB_CHUNK_F);
#pragma unroll
        for (int i = 0; i < 4; ++i) dstv[tid + 256 * i] = stg[i];
        if (tid < CHUNK_V4 - 1024) dstv[tid + 1024] = stg[4];
    };
    stage_load(0);
    __syncthreads();
    {   // relu(conv1) rows of the tile, zero outside [0, L)
        const int cq = tid & 15, rl = tid >> 4;
        float w1[4][9], b1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int d = 0; d < 9; ++d) w1[i][d] = p.c1[(4 * cq + i) * 10 + d];
            b1[i] = p.c1[(4 * cq + i) * 10 + 9];
        }
        for (int row = rl; row < TR; row += 16) {
            const int t = tbase + row;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = b1[i];
#pragma unroll
                for (int d = 0; d < 9; ++d) a = fmaf(w1[i][d], raw[row + d], a);
                v[i] = fmaxf(a, 0.f);
            }
            const bool valid = (t >= 0) && (t < L);
            st4(act + row * ROWF + 4 * cq,
                valid ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    stage_store(0);
    __syncthreads();

    int c = 0;
    for (int ocb = 0; ocb < 4; ++ocb) {
        floatx16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
        for (int cc = 0; cc < 10; ++cc, ++c) {
            const int d = cc >> 1, hh = cc & 1;
            if (c + 1 < SGB_NCHUNK) stage_load(c + 1);
            const float* wb = wbuf + (c & 1) * SGB_CHUNK_F + (32 * wave + ln) * WROWF + 4 * lh;
            const float* arow = act + (ln + d) * ROWF + 32 * hh + 4 * lh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = ld4(wb + 8 * q);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float4 a = ld4(arow + 32 * m * ROWF + 8 * q);
                    acc[m] = mfma4(a, b, acc[m]);
                }
            }
            if (cc == 9) {
                // pool: accumulator register v of M-tile m is time row 32m + (v&3) + 8(v>>2) + 4*lh,
                // so an 8-row register group never straddles a window of 80
                const int oc = 128 * ocb + 32 * wave + ln;
                const float bias = p.cbias[oc];
                float wmax[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) wmax[w] = -INFINITY;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int w = (32 * m + 8 * (v >> 2)) / SGB_SCALE;
                        wmax[w] = fmaxf(wmax[w], acc[m][v]);
                    }
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    float mval = fmaxf(wmax[w], __shfl_xor(wmax[w], 32));
                    mval += bias;
                    mval = mval > 0.f ? mval : 0.01f * mval;
                    if (lh == 0 && w0 + w < p.P)
                        p.pooled[((size_t)n * p.P + w0 + w) * NF_SGB + oc] = mval;
                }
            }
            if (c + 1 < SGB_NCHUNK) stage_store((c + 1) & 1);
            __syncthreads();
        }
    }
}

// expand_conv 512->64 k5 on the pooled grid + lrelu (0.2 % of the FLOPs): plain fp32 FMA.
// One work-group = 16 pooled columns of one waveform; thread = (oc, group of 4 columns).
constexpr int EXP_COLS = 16;
__global__ __launch_bounds__(256) void sgb_expand_kernel(const float* __restrict__ pooled,
                                                         const float* __restrict__ ew,
                                                         const float* __restrict__ ebias,
                                                         float* __restrict__ sgb, int N, int P,
                                                         int blocks_per_wf) {
    __shared__ __attribute__((aligned(16))) float tile[(EXP_COLS + 4) * NF_SGB];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / blocks_per_wf;
    const int wbase = (blockIdx.x - n * blocks_per_wf) * EXP_COLS;
    for (int i = tid; i < (EXP_COLS + 4) * NF_SGB / 4; i += 256) {
        const int col = i / (NF_SGB / 4), c4 = i - col * (NF_SGB / 4);
        const int w = wbase - 2 + col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w >= 0 && w < P) v = ld4(pooled + ((size_t)n * P + w) * NF_SGB + 4 * c4);
        st4(tile + col * NF_SGB + 4 * c4, v);
    }
    __syncthreads();
    const int oc = tid & 63, cg = tid >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < 5; ++d) {
        const float* wp = ew + (size_t)d * NF_SGB * NF + oc;
        const float* tp = tile + (4 * cg + d) * NF_SGB;
#pragma unroll 4
        for (int ch = 0; ch < NF_SGB; ++ch) {
            const float w = wp[(size_t)ch * NF];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(w, tp[k * NF_SGB + ch], acc[k]);
        }
    }
    const float b = ebias[oc];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int w = wbase + 4 * cg + k;
        if (w < P) {
            float v = acc[k] + b;
            v = v > 0.f ? v : 0.01f * v;
            sgb[((size_t)n * P + w) * NF + oc] = v;
        }
    }
}

constexpr int BODY_S = 192, BODY_RING = 256, BODY_RAWRING = 256;
constexpr int SGB_NW = 4;
constexpr int64_t SUB_BATCH = 4096;      // rows whose SGB maps share one workspace

size_t sgb_lds_bytes() {
    return (size_t)((SGB_SCALE * SGB_NW + 4) * ROWF + 2 * SGB_CHUNK_F + SGB_SCALE * SGB_NW + 12) * sizeof(float);
}

}  // namespace

extern "C" size_t stof_forward_workspace_bytes(const stof_net_desc* desc, int64_t N, int64_t L) {
    if (!desc || N <= 0 || L <= 0 || desc->semi_global_scale == 1) return 0;
    const int64_t nb = N < SUB_BATCH ? N : SUB_BATCH;
    const int64_t P = L / SGB_SCALE;
    return (size_t)(nb * P * (NF_SGB + NF)) * sizeof(float) + 256;
}

static int forward_impl(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y,
                        int64_t N, int64_t L, void* workspace, size_t workspace_bytes, void* stream_,
                        void* const* events) {
    if (!desc || N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    if ((N == 0 || L == 0) && desc->precision == STOF_PREC_FP32) return STOF_OK;   // empty batch: nothing to do
    if (!packed_dev || !x || !y) return STOF_ERR_BAD_ARG;
    if (desc->precision != STOF_PREC_FP32) return STOF_ERR_UNSUPPORTED;
    const int r = desc->upsample_factor;
    if (r < 1 || r > 64) return STOF_ERR_UNSUPPORTED;
    const bool has_sgb = desc->semi_global_scale != 1;
    if (has_sgb && desc->semi_global_scale != SGB_SCALE) return STOF_ERR_UNSUPPORTED;
    if (N == 0 || L == 0) return STOF_OK;
    const int64_t P = L / SGB_SCALE;
    const int64_t rem = L - P * SGB_SCALE;
    if (has_sgb && (rem & 1)) return STOF_ERR_ODD_SGB_REMAINDER;
    if ((L + GAP) * SUB_BATCH > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;   // local stream rows are int32
    if (has_sgb && (!workspace || workspace_bytes < stof_forward_workspace_bytes(desc, N, L)))
        return STOF_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    // layout of the packed blob (mirrors pack_weights.cpp)
    const float* base = static_cast<const float*>(packed_dev);
    uint64_t off = sizeof(PackedHeader) / sizeof(float);
    const float* c1 = base + off;      off += 64 * 10;
    const float* bias = base + off;    off += 13 * 64;
    const float* body = base + off;    off += (uint64_t)BODY_NCHUNK * BODY_CHUNK_F;
    const float* cbias = base + off;   off += NF_SGB;
    const float* cchunks = base + off; off += (uint64_t)SGB_NCHUNK * SGB_CHUNK_F;
    const float* ew = base + off;      off += 5ull * NF_SGB * NF;
    const float* ebias = base + off;

    using Lds = BodyLds<BODY_S, BODY_RING, BODY_RAWRING>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&body_sweep_kernel<BODY_S, BODY_RING, BODY_RAWRING>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lds::BYTES) != hipSuccess)
            return STOF_ERR_HIP;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sgb_contract_pool_kernel<SGB_NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sgb_lds_bytes()) != hipSuccess)
            return STOF_ERR_HIP;
        attr_done = true;
    }
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }

    for (int64_t b0 = 0; b0 < N; b0 += SUB_BATCH) {
        const int64_t nb = (N - b0) < SUB_BATCH ? (N - b0) : SUB_BATCH;
        const float* xb = x + b0 * L;
        float* yb = y + b0 * L * r;
        float* pooled = nullptr;
        float* sgb = nullptr;
        if (has_sgb && P > 0) {
            pooled = static_cast<float*>(workspace);
            sgb = pooled + nb * P * NF_SGB;
            SgbParams sp;
            sp.x = xb; sp.pooled = pooled; sp.c1 = c1; sp.cbias = cbias; sp.chunks = cchunks;
            sp.N = (int)nb; sp.L = (int)L; sp.P = (int)P;
            sp.tiles_per_wf = (int)((P + SGB_NW - 1) / SGB_NW);
            if (events && b0 == 0) (void)hipEventRecord(static_cast<hipEvent_t>(events[0]), stream);
            hipLaunchKernelGGL(sgb_contract_pool_kernel<SGB_NW>, dim3((unsigned)(nb * sp.tiles_per_wf)), dim3(256),
                               sgb_lds_bytes(), stream, sp);
            if (events && b0 == 0) (void)hipEventRecord(static_cast<hipEvent_t>(events[1]), stream);
            const int bpw = (int)((P + EXP_COLS - 1) / EXP_COLS);
            hipLaunchKernelGGL(sgb_expand_kernel, dim3((unsigned)(nb * bpw)), dim3(256), 0, stream,
                               pooled, ew, ebias, sgb, (int)nb, (int)P, bpw);
            if (events && b0 == 0) (void)hipEventRecord(static_cast<hipEvent_t>(events[2]), stream);
        } else if (events && b0 == 0) {
            for (int e = 0; e < 3; ++e) (void)hipEventRecord(static_cast<hipEvent_t>(events[e]), stream);
        }
        BodyParams bp;
        bp.x = xb; bp.sgb = (has_sgb && P > 0) ? sgb : nullptr; bp.y = yb;
        bp.c1 = c1; bp.bias = bias; bp.chunks = body;
        bp.N = (int)nb; bp.L = (int)L; bp.r = r; bp.P = (int)P; bp.rem_half = (int)(rem / 2);
        // one persistent work-group per CU, each sweeping a contiguous run of waveforms
        int64_t wgs = nb < ncu ? nb : ncu;
        bp.wf_per_wg = (int)((nb + wgs - 1) / wgs);
        wgs = (nb + bp.wf_per_wg - 1) / bp.wf_per_wg;
        hipLaunchKernelGGL((body_sweep_kernel<BODY_S, BODY_RING, BODY_RAWRING>), dim3((unsigned)wgs), dim3(256),
                           Lds::BYTES, stream, bp);
        if (events && b0 == 0) (void)hipEventRecord(static_cast<hipEvent_t>(events[3]), stream);
    }
    if (hipGetLastError() != hipSuccess) return STOF_ERR_HIP;
    return STOF_OK;
}

extern "C" int stof_forward(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y,
                            int64_t N, int64_t L, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_impl(desc, packed_dev, x, y, N, L, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int stof_forward_events(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y,
                                   int64_t N, int64_t L, void* workspace, size_t workspace_bytes, void* stream,
                                   void* const* events) {
    if (!events) return STOF_ERR_BAD_ARG;
    for (int e = 0; e < STOF_FORWARD_EVENTS; ++e)
        if (!events[e]) return STOF_ERR_BAD_ARG;
    return forward_impl(desc, packed_dev, x, y, N, L, workspace, workspace_bytes, stream, events);
}

extern "C" int stof_events_create(int32_t count, void** events_out) {
    if (count < 0 || !events_out) return STOF_ERR_BAD_ARG;
    for (int i = 0; i < count; ++i) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return STOF_ERR_HIP;
        events_out[i] = ev;
    }
    return STOF_OK;
}

extern "C" int stof_events_destroy(int32_t count, void* const* events) {
    if (count < 0 || !events) return STOF_ERR_BAD_ARG;
    for (int i = 0; i < count; ++i)
        if (events[i]) (void)hipEventDestroy(static_cast<hipEvent_t>(events[i]));
    return STOF_OK;
}

extern "C" int stof_event_elapsed_ms(void* start, void* stop, float* ms_out) {
    if (!start || !stop || !ms_out) return STOF_ERR_BAD_ARG;
    return hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) == hipSuccess
               ? STOF_OK : STOF_ERR_HIP;
}
