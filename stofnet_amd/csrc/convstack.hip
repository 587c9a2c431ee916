// StofNet.forward (models/stofnet.py:42-67) as three gfx950 kernels:
//
//   sgb_contract_pool : x -> relu(conv1) -> contract_conv 64->512 k5 (MFMA) -> lrelu
//                       -> max-pool 80  => pooled[N][P][512]        (models/stofnet.py:45,100-103)
//   expand (train.hip conv_cl_kernel, stream mode): pooled -> expand_conv 512->64 k5 -> lrelu => sgb[N][P][64] (:106-107)
//   body_sweep        : x, sgb -> relu(conv1) + upsample/pad(sgb) (:45,108-115) -> conv2..conv12
//                       with the residual pattern of :51-62 -> conv_last -> SampleShuffle1D store
//                       (:65, utils/sample_shuffle.py:10-28)        => y[N][L*r]
//
// All activations between conv1 and the shuffle store live in LDS: body_sweep walks a
// stream of waveforms left to right in steps of S rows; every layer keeps a frontier that
// lags 3 rows per conv behind the previous one, the residual stream lives in ring X
// (updated in place), intermediates in ring Y.  No halo is recomputed and nothing but x,
// sgb and y touches HBM.  The schedule is emulated in numpy by oracle/sweep_emulator.py.
//
// Weights never pass through LDS: they are packed in MFMA-fragment order (stof_common.h) and
// each wave pulls its next operand fragments from L2 with coalesced 1-KiB loads two chunks
// ahead of use, so a layer needs one work-group barrier, not one per weight tile.
//
// Two arithmetic modes (template PREC):
//   STOF_PREC_FP32  exact fp32 on v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain): parity baseline
//   STOF_PREC_F16X3 operands split x = hi + lo in fp16 (|err| ~ 2^-22 |x|), three
//                   v_mfma_f32_32x32x16_f16 passes hi*hi + hi*lo + lo*hi, fp32 accumulate
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>
#include "stof_common.h"
#include "stof_hip_util.h"

// expand_conv 512->64 k5 on the pooled grid + lrelu runs on the channel-last MFMA conv of train.hip in stream mode (the P
// pooled columns of every waveform followed by 2 zero gap rows form one long row sequence, so the 128-row tiles are full).
namespace stof {
int launch_conv_cl(const float* x, const float* w, const float* bias, const float* residual, const float* saved, float* y,
                   int64_t N, int64_t L, int32_t cin, int32_t cout, int32_t K, int32_t act, int32_t precision,
                   int32_t period, int32_t valid_len, hipStream_t stream, const int* run_if);
}

using namespace stof;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LAG_LAST = 34;      // frontier lag of conv_last: 11 convs x 3 + 1
#ifndef STOF_SGB_NW
#define STOF_SGB_NW 2          // measured: 2-window tiles with two work-groups per CU beat 4-window tiles by 8 %
#endif
constexpr int SGB_NW = STOF_SGB_NW;                    // pooling windows per work-group tile
constexpr int SGB_WAVES_PER_SIMD = (SGB_NW <= 2) ? 2 : 1;   // small tiles: two work-groups share a CU
constexpr int ROWB = ROWF * 4;    // activation row stride in bytes

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int layer_lag(int j) { return j <= 11 ? 3 * j : LAG_LAST; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ uint4 ldq(const char* p) { return *reinterpret_cast<const uint4*>(p); }

__device__ __forceinline__ float4 as_f4(uint4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ half8 as_h8(uint4 v) {
    union { uint4 u; half8 h; } c;
    c.u = v;
    return c.h;
}

// D += A * B over 8 channels in exact fp32: lane (i, h) of both operands holds channels
// 4h..4h+3 of the group, step s contracts channels {s, 4+s}.
__device__ __forceinline__ floatx16 mma_fp32(uint4 a, uint4 b, floatx16 c) {
    const float4 af = as_f4(a), bf = as_f4(b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, c, 0, 0, 0);
    return c;
}
// D += (Ah + Al) * (Bh + Bl) without the lo*lo term, over 16 channels.  (Keeping the two cross
// terms in an accumulator of their own was measured: no accuracy gain worth its 12 % slowdown.)
__device__ __forceinline__ floatx16 mma_f16x3(uint4 ah, uint4 al, uint4 bh, uint4 bl, floatx16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah), as_h8(bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah), as_h8(bl), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(al), as_h8(bh), c, 0, 0, 0);
    return c;
}

// Byte offset inside an activation row of operand fragment `frag` (see stof_common.h) for the
// 32-channel half hh and lane half lh.
template <int PREC>
__device__ __forceinline__ int act_frag_off(int frag, int hh, int lh) {
    if constexpr (PREC == STOF_PREC_FP32) return (32 * hh + 8 * frag + 4 * lh) * 4;
    else return (32 * hh + 16 * (frag >> 1) + 8 * lh) * 2 + 128 * (frag & 1);
}

// Store 4 consecutive channels c0..c0+3 of one activation row (row = row base pointer).
template <int PREC>
__device__ __forceinline__ void store_act4(char* row, int c0, float4 v) {
    if constexpr (PREC == STOF_PREC_FP32) {
        *reinterpret_cast<float4*>(row + 4 * c0) = v;
    } else {
        half4 hi, lo;
        hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
        lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
        lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
        *reinterpret_cast<half4*>(row + 2 * c0) = hi;
        *reinterpret_cast<half4*>(row + 128 + 2 * c0) = lo;
    }
}
template <int PREC>
__device__ __forceinline__ float4 load_act4(const char* row, int c0) {
    if constexpr (PREC == STOF_PREC_FP32) {
        return *reinterpret_cast<const float4*>(row + 4 * c0);
    } else {
        const half4 hi = *reinterpret_cast<const half4*>(row + 2 * c0);
        const half4 lo = *reinterpret_cast<const half4*>(row + 128 + 2 * c0);
        return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1],
                           (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]);
    }
}

// v + (float)h and v - (float)h as ONE mixed-precision FMA each (v_fma_mix_f32: fp16 source, fp32 arithmetic).  `one` is a
// 1.0f the optimiser cannot see through (opaque_one()), so the expression stays an FMA with an fpext operand, which the
// backend selects as v_fma_mix_f32; this file is built with -fno-slp-vectorize because the SLP vectoriser would otherwise pair
// two of them into v_cvt_f32_f16 x2 + v_pk_fma_f32, and packed-f32 VALU is slow beside MFMAs (MI355X_MICROARCH.md).
__device__ __forceinline__ float opaque_one() {
    float one;
    asm volatile("s_mov_b32 %0, 1.0" : "=s"(one));
    return one;
}
__device__ __forceinline__ float mix_add(_Float16 h, float v, float one) { return __builtin_fmaf((float)h, one, v); }
__device__ __forceinline__ float mix_sub(_Float16 h, float v, float one) { return __builtin_fmaf(-(float)h, one, v); }
__device__ __forceinline__ half2v cvt_h2(float a, float b) {
    const float2v v = {a, b};
    return __builtin_convertvector(v, half2v);         // v_cvt_pk_f16_f32 (round to nearest even)
}
__device__ __forceinline__ unsigned h2_bits(half2v h) {
    union { half2v h; unsigned u; } c;
    c.h = h;
    return c.u;
}
__device__ __forceinline__ half2v bits_h2(unsigned u) {
    union { half2v h; unsigned u; } c;
    c.u = u;
    return c.h;
}

#ifdef STOF_STAMPS
// diagnostic build: shader-clock stamp with its own wait (cdna_hip_programming.md section 7)
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP_ADD(slot) do { const unsigned long long t_ = stamp(); tsum[slot] += t_ - tprev; tprev = t_; } while (0)
#else
#define STAMP_ADD(slot) do {} while (0)
#endif

// ----------------------------------------------------------------------------------
// body sweep
// ----------------------------------------------------------------------------------
// What a 16-row tile of the network output (16 time rows x r <= 16 channels = 16 r consecutive-ish map positions of one
// waveform) contributes to get_maxima_positions in arg-max mode (utils/mask2samples.py:14-34): its maximum, its
// minimum and the exact set of positions that equal its maximum -- bit (e, lane) of eq[e] is output channel
// 4 (lane >> 4) + e of time row tw_base + (lane & 15), i.e. map position (tw_base + (lane & 15)) * r + 4 (lane >> 4) + e.
struct OnsetPartial {
    int written;                  // 0 = slot not used
    int tw_base;
    float m, lo;
    unsigned long long eq[4];
};
static_assert(sizeof(OnsetPartial) == 48, "OnsetPartial is three 16-byte stores");

struct BodyParams {
    const float* x;        // [N][L]
    const float* sgb;      // [N][P][64] or nullptr
    float* y;              // [N][L*r]
    const float* c1;       // [64][10]
    const float* bias;     // [13][64]
    const float* chunks;   // [BODY_NCHUNK][BODY_CHUNK_F] fragment-ordered weights
    const float* last16;   // f16x3, r <= 16: conv_last as 16x16x32 MFMA operands (stof_common.h), else nullptr
    int N, L, r, P, rem_half, wf_per_wg;
    // Small batches: every waveform is cut into 2^nseg_log2 segments of seg_len rows that are swept as independent
    // 'virtual waveforms' with `halo` real rows of context on both sides (the stack's receptive field is +-38), so
    // that the CUs are not left idle.  N counts virtual waveforms; nseg_log2 = 0, seg_len = L, halo = 0 is the plain case.
    int nseg_log2, seg_len, halo;
    unsigned long long* stamps;   // diagnostic builds (-DSTOF_STAMPS) only: [wg][wave][8] cycle sums
    int* status;                  // optional: bit 0 set if a non-finite output was produced (f16x3 range overflow)
    const int* run_if;            // optional: the whole launch returns at once while *run_if == 0 ('auto' precision re-run)
    // fused arg-max picker (stof_forward_onsets): every 16-row tile of conv_last's output leaves one OnsetPartial per
    // (virtual) waveform it touches; onsets_finalize_kernel turns them into counts / indices.  y may then be nullptr.
    OnsetPartial* onset_ws;       // [N][onset_slots] (this sub-batch), zeroed by the host side before the launch
    int onset_slots, onset_seg_slots;
    // training forward (stof_train_sweep, 16x16x32 body only): every layer's output is also written to HBM, channel-last fp32
    // [12][N][L][64]: tensor 0 = x0 (relu(conv1) + SemiGlobalBlock), 1..10 = outputs of conv2..conv11 (after leaky ReLU /
    // residual add), 11 = conv12's; followed by >= 2 KiB the kernel may scribble on (rows that are padding)
    float* dump;
    long long dump_stride;        // floats per tensor = N * L * 64
    // backward sweep: gin = gradient at conv12's output, channel-last [N][L][64]; fwd_dump = the forward sweep's dump, whose
    // tensors 1, 3, .., 9 (outputs of conv2, 4, .., 10 after their leaky ReLU) give the leaky-ReLU derivatives by their sign
    const float* gin;
    const float* fwd_dump;
};

// RF = floats per activation row: ROWF (272 B) for the 32-wide MFMA shapes; 72 (288 B) for the 16x16x32 body, whose
// operand reads (lane = (time row i, k-group q), 16 B at 16 q) are bank-conflict free exactly when the row stride is
// 2 mod 4 in 16-byte units: unit index mod 16 = 2 i + q + const, and the hardware's ds_read_b128 lane groups pair rows
// {0-3, 12-15} of one q with rows {4-11} of q + 1 (MI355X_MICROARCH.md, LDS table) -- even values against odd ones.
template <int S, int RING, int RAWRING, int RF = ROWF>
struct BodyLds {
    static constexpr int X = 0;
    static constexpr int Y = X + RING * RF;
    static constexpr int RAW = Y + RING * RF;
    static constexpr int BIAS = RAW + RAWRING;
    static constexpr int SGL = BIAS + 13 * 64;           // [8 windows][64 ch] SemiGlobalBlock rows
    static constexpr int TOTAL = SGL + 8 * 64;
    static constexpr size_t BYTES = (size_t)TOTAL * sizeof(float);
};

constexpr int ROWF16 = 72;        // 288-byte rows of the 16x16x32 body (see BodyLds)

// BWD: the data-gradient chain of the training step as the same sweep run backwards through the network (stof_train_sweep_bwd):
// "layer 0" loads dL/dx6 rows from HBM instead of computing conv1, sweep layer j = 1..11 is the transposed convolution of
// conv(13 - j): j = 1 plain, even j times the leaky-ReLU derivative of the saved activation (sign bytes), odd j >= 3 added to
// the residual gradient in place; every layer's output goes to HBM for the weight-gradient kernels (DUMP); no conv_last.
template <int PREC, int S, int RING, int RAWRING, int SHAPE = 32, bool DUMP = false, bool BWD = false>
__global__ __launch_bounds__(256, 1) void body_sweep_kernel(const BodyParams p) {
    static_assert(!DUMP || SHAPE == 16, "the training dump lives in the 16x16x32 body");
    static_assert(!BWD || DUMP, "the backward sweep's outputs are its dumps");
    constexpr int NCHUNK_STEP = BWD ? 11 * BODY_CHUNKS_K7 : BODY_NCHUNK;      // weight chunks consumed per sweep step
    static_assert((RING & (RING - 1)) == 0 && (RAWRING & (RAWRING - 1)) == 0, "rings are powers of two");
    static_assert(S % 64 == 0 && S + 36 <= RING && S + 42 <= RAWRING, "ring must hold the live span");
    static_assert(SHAPE == 32 || (SHAPE == 16 && PREC == STOF_PREC_F16X3), "the 16x16x32 body is a split-fp16 kernel");
    constexpr int RF = SHAPE == 16 ? ROWF16 : ROWF;
    constexpr int ROWB = RF * 4;                 // activation row stride in bytes (shadows the namespace constant)
    using Lds = BodyLds<S, RING, RAWRING, RF>;
    constexpr int NT = S / 64;                   // N-tiles (32 rows) per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const Xr = reinterpret_cast<char*>(smem + Lds::X);
    char* const Yr = reinterpret_cast<char*>(smem + Lds::Y);
    float* const rawr = smem + Lds::RAW;
    float* const biasl = smem + Lds::BIAS;
    float* const sgl = smem + Lds::SGL;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave & 1, ni = wave >> 1;
    const int ln = lane & 31, lh = lane >> 5;

    const int n0 = blockIdx.x * p.wf_per_wg;
    const int n1 = min(p.N, n0 + p.wf_per_wg);
    if (n0 >= n1) return;
    if (p.run_if != nullptr && *p.run_if == 0) return;
    const int Ltrue = p.L, r = p.r;
    const int L = p.seg_len + 2 * p.halo;        // rows of one (virtual) waveform in the stream
    const int Lp = L + GAP;
    const int gend = (n1 - n0) * Lp;             // local stream rows [0, gend)
    const int seg_mask = (1 << p.nseg_log2) - 1;
    const bool seg_mode = p.nseg_log2 > 0;
    // (virtual waveform, local row) -> (waveform, time); rows whose time falls outside [0, Ltrue) are padding
    auto vmap = [&](int nv, int tl, int& n, int& tt) {
        n = nv >> p.nseg_log2;
        tt = (nv & seg_mask) * p.seg_len - p.halo + tl;
    };

    // ---- one-time setup: zero rings, biases to LDS, conv1 taps to registers
    for (int i = tid; i < Lds::RAW + RAWRING; i += 256) smem[i] = 0.f;
    for (int i = tid; i < 13 * 64; i += 256) biasl[i] = p.bias[i];
    const int cq = tid & 15, rl = tid >> 4;
    float w1[4][9], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int d = 0; d < 9; ++d) w1[i][d] = p.c1[(4 * cq + i) * 10 + d];
        b1[i] = p.c1[(4 * cq + i) * 10 + 9];
    }
    __syncthreads();
    // Row decode without division: (n, t) of stream row (base row + off), given the base row's
    // (nB, tB); off < S + 64.  One conditional subtraction when Lp >= the offset range, a short
    // loop otherwise (tiny waveforms).
    auto decode_row = [&](int nB, int tB, int off, int& n, int& t) {
        n = nB;
        t = tB + off;
        if (Lp >= 2 * S) {
            if (t >= Lp) { t -= Lp; n += 1; }
        } else {
            while (t >= Lp) { t -= Lp; n += 1; }
        }
    };

    // relu(conv1(x)) + SemiGlobalBlock contribution for stream rows [rstart, rstart+S) -> ring dst.
    // A thread owns 4 channels x NIT consecutive rows: the NIT+8 raw samples it needs are read from
    // LDS once, rows are decoded incrementally (one division per pass), and every SemiGlobalBlock
    // load is issued before the first FMA so L2 latency is paid once per pass.
    // (nR, tR) = waveform / time of stream row `rstart`, maintained incrementally by the caller
    const float one_x0 = opaque_one();
    auto x0_pass = [&](char* dst, int rstart, int nR, int tR, bool dump_it) {
        constexpr int NIT = S / 16;
        float* const dump0 = (DUMP && dump_it) ? p.dump : nullptr;
        const int g0 = rstart + rl * NIT;
        // Fast path (work-group uniform test): the pass's S rows lie inside ONE waveform, all of them real samples and (with a
        // SemiGlobalBlock) inside the up-sampled map -- the common case by far (a waveform is ~10 steps long).  Then no row
        // needs decoding or masking, and a thread's NIT consecutive rows meet at most two pooling windows, whose rows are
        // read from LDS once.  The general path below handles waveform boundaries, padding rows and the segment mode.
        if (!seg_mode && rstart >= 0 && rstart + S <= gend && tR + S <= L &&
            (p.sgb == nullptr || (tR >= p.rem_half && tR + S - p.rem_half <= SGB_SCALE * p.P))) {
            float xs[NIT + 8];
#pragma unroll
            for (int i = 0; i < NIT + 8; ++i) xs[i] = rawr[(g0 - 4 + i) & (RAWRING - 1)];
            float4 sgA = make_float4(0.f, 0.f, 0.f, 0.f), sgB = sgA;
            int sw = NIT;                                   // rows of this thread that belong to its first window
            if (p.sgb != nullptr) {
                const int pos0 = tR + rl * NIT - p.rem_half;
                const int w0 = pos0 / SGB_SCALE;
                sw = SGB_SCALE * (w0 + 1) - pos0;
                const int wid = (n0 + nR) * p.P + w0;
                sgA = ld4(sgl + (wid & 7) * NF + 4 * cq);
                sgB = ld4(sgl + ((wid + 1) & 7) * NF + 4 * cq);   // staged only if a row of the pass meets it; unused otherwise
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const bool first = it < sw;
                const float sg[4] = {first ? sgA.x : sgB.x, first ? sgA.y : sgB.y, first ? sgA.z : sgB.z, first ? sgA.w : sgB.w};
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = b1[i];
#pragma unroll
                    for (int d = 0; d < 9; ++d) a = fmaf(w1[i][d], xs[it + d], a);
                    v[i] = fmaxf(a, 0.f) + sg[i];
                }
                char* const row = dst + ((g0 + it) & (RING - 1)) * ROWB;
                if (dump0 != nullptr)
                    st4(dump0 + ((size_t)(n0 + nR) * Ltrue + tR + rl * NIT + it) * NF + 4 * cq, make_float4(v[0], v[1], v[2], v[3]));
                if constexpr (PREC == STOF_PREC_FP32) {
                    *reinterpret_cast<float4*>(row + 16 * cq) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    const half2v h01 = cvt_h2(v[0], v[1]), h23 = cvt_h2(v[2], v[3]);
                    const half2v l01 = cvt_h2(mix_sub(h01[0], v[0], one_x0), mix_sub(h01[1], v[1], one_x0));
                    const half2v l23 = cvt_h2(mix_sub(h23[0], v[2], one_x0), mix_sub(h23[1], v[3], one_x0));
                    *reinterpret_cast<uint2*>(row + 8 * cq) = make_uint2(h2_bits(h01), h2_bits(h23));
                    *reinterpret_cast<uint2*>(row + 128 + 8 * cq) = make_uint2(h2_bits(l01), h2_bits(l23));
                }
            }
            return;
        }
        int nb, tb;
        decode_row(nR, tR, rl * NIT, nb, tb);
        float xs[NIT + 8];
#pragma unroll
        for (int i = 0; i < NIT + 8; ++i) xs[i] = rawr[(g0 - 4 + i) & (RAWRING - 1)];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int g = g0 + it;
            int t = tb + it;
            int nl = nb;
            if (t >= Lp) { t -= Lp; nl += 1; }             // NIT < Lp: at most one wrap
            int nw, tw;
            vmap(n0 + nl, t, nw, tw);
            const bool ok = (g >= 0) && (g < gend) && (t < L) && (tw >= 0) && (tw < Ltrue);
            float4 sg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.sgb != nullptr && ok) {
                const int pos = tw - p.rem_half;
                if (pos >= 0 && pos < SGB_SCALE * p.P) {
                    const int wid = nw * p.P + pos / SGB_SCALE;             // global window id
                    // staged in LDS by the step prologue; segments start anywhere inside a window, which the
                    // prologue's probe rows do not cover, so the (throughput-insensitive) segment mode reads L2
                    sg = seg_mode ? ld4(p.sgb + (size_t)wid * NF + 4 * cq) : ld4(sgl + (wid & 7) * NF + 4 * cq);
                }
            }
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = b1[i];
#pragma unroll
                for (int d = 0; d < 9; ++d) a = fmaf(w1[i][d], xs[it + d], a);
                v[i] = fmaxf(a, 0.f);
            }
            const float4 o = ok ? make_float4(v[0] + sg.x, v[1] + sg.y, v[2] + sg.z, v[3] + sg.w)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
            if (dump0 != nullptr && ok && t >= p.halo && t < p.halo + p.seg_len)        // a segment's own rows only
                st4(dump0 + ((size_t)nw * Ltrue + tw) * NF + 4 * cq, o);
            store_act4<PREC>(dst + ((g0 + it) & (RING - 1)) * ROWB, 4 * cq, o);
        }
    };

    // Per-step global fetches, issued one step ahead: thread tid < S owns raw row Fn+4-S+tid; the
    // four probe rows Fn-S + {0, 80, 160, S-1} hit every 80-sample window that meets [Fn-S, Fn), and
    // the 64 threads of probe q fetch that window's 64 channels.  (nB, tB) decode row Fn - S.
    float raw_next = 0.f, sg_next = 0.f;
    int sg_slot = -1;
    auto fetch_step = [&](int Fn, int nB, int tB) {
        raw_next = 0.f;
        if (tid < S) {
            const int g = Fn + 4 - S + tid;
            int nl, t;
            decode_row(nB, tB, 4 + tid, nl, t);
            int nw, tw;
            vmap(n0 + nl, t, nw, tw);
            if (g < gend && t < L && tw >= 0 && tw < Ltrue) raw_next = p.x[(size_t)nw * Ltrue + tw];
        }
        sg_slot = -1;
        if (p.sgb != nullptr && !seg_mode) {
            const int q = tid >> 6;
            const int off = q == 3 ? S - 1 : 80 * q;
            const int g = Fn - S + off;
            int nl, t;
            decode_row(nB, tB, off, nl, t);
            const int pos = t - p.rem_half;
            if (g < gend && t < L && pos >= 0 && pos < SGB_SCALE * p.P) {
                const int wid = (n0 + nl) * p.P + pos / SGB_SCALE;
                sg_slot = wid & 7;
                sg_next = p.sgb[(size_t)wid * NF + (tid & 63)];
            }
        }
    };
    static_assert(S <= 240 && S > 160, "probe offsets {0, 80, 160, S-1} assume 160 < S <= 240");
    if constexpr (!BWD) {
        if (tid < 4 && tid < L) {                                        // rows 0..3 precede the first fetch
            int nw, tw;
            vmap(n0, tid, nw, tw);
            if (tw >= 0 && tw < Ltrue) rawr[tid] = p.x[(size_t)nw * Ltrue + tw];
        }
        fetch_step(S, 0, 0);
    }
    // backward sweep, "layer 0": stream rows [rstart, rstart + S) of dL/dx6 (channel-last fp32 in HBM) into ring dst as split fp16;
    // same thread mapping as x0_pass (4 channels x NIT consecutive rows), all loads issued before the first store
    auto gin_pass = [&](char* dst, int rstart, int nR, int tR) {
        constexpr int NIT = S / 16;
        const int g0 = rstart + rl * NIT;
        int nb, tb;
        decode_row(nR, tR, rl * NIT, nb, tb);
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int g = g0 + it;
            int t = tb + it, nl = nb;
            if (t >= Lp) { t -= Lp; nl += 1; }
            int nw, tw;
            vmap(n0 + nl, t, nw, tw);
            const bool ok = (g >= 0) && (g < gend) && (t < L) && (tw >= 0) && (tw < Ltrue);
            v[it] = ok ? ld4(p.gin + ((size_t)nw * Ltrue + tw) * NF + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) store_act4<PREC>(dst + ((g0 + it) & (RING - 1)) * ROWB, 4 * cq, v[it]);
    };

    // ---- weight fragments: registers, fetched two chunks ahead of use
    const uint4* const wbase = reinterpret_cast<const uint4*>(p.chunks) + mi * 64 + lane;
    auto wload = [&](int c, int f) -> uint4 { return wbase[((size_t)c * FRAGS_PER_CHUNK + f) * 128]; };
    uint4 wf[2][FRAGS_PER_CHUNK];
#pragma unroll
    for (int f = 0; f < FRAGS_PER_CHUNK; ++f) {
        wf[0][f] = wload(0, f);
        wf[1][f] = wload(1, f);
    }

#ifdef STOF_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = stamp();
    const unsigned long long tstart = tprev;
#endif
    const int nsteps = (gend - GAP + LAG_LAST + S - 1) / S;
    int nS = 0, tS = 0;                           // waveform / time of stream row F - S (wave-uniform)
    for (int step = 1; step <= nsteps; ++step) {
        const int F = step * S;
        if (step > 1) {
            tS += S;
            while (tS >= Lp) { tS -= Lp; nS += 1; }
        }
        // The raw samples [F+4-S, F+4) and the SemiGlobalBlock rows of the windows that meet
        // [F-S, F) were fetched into registers one step ago (or by the prologue): land them in LDS,
        // then start the fetch for the next step so its latency hides behind this step's MFMAs.
        if constexpr (!BWD) {
            if (tid < S) rawr[(F + 4 - S + tid) & (RAWRING - 1)] = raw_next;
            if (sg_slot >= 0) sgl[sg_slot * NF + (tid & 63)] = sg_next;
            int nN = nS, tN = tS + S;                     // decode of row F (first row of the next step)
            while (tN >= Lp) { tN -= Lp; nN += 1; }
            fetch_step(F + S, nN, tN);
        }
        __syncthreads();
        STAMP_ADD(0);                             // raw load + barrier
        if constexpr (BWD) gin_pass(Xr, F - S, nS, tS);
        else x0_pass(Xr, F - S, nS, tS, true);    // sweep layer 0
        STAMP_ADD(1);                             // x0 passes
        __syncthreads();
        STAMP_ADD(2);                             // barrier waits

        int c = 0;                                // chunk index within the step
        for (int j = 1; j <= (BWD ? 11 : 12); ++j) {
            // waveform / time of the layer's first row R0 = F - S - lag (may precede the stream: n = -1)
            int nR = nS, tR = tS - layer_lag(j);
            while (tR < 0) { tR += Lp; nR -= 1; }
            if (!BWD && j == 11) {                // long skip: seed the destination with x0
                x0_pass(Yr, F - S - 33, nR, tR, false);
                STAMP_ADD(1);
                __syncthreads();
                STAMP_ADD(2);
            }
            const bool last = (j == 12);
            const bool reads_x = (j & 1);         // conv2,4,..,10 and conv12 read ring X
            const char* const src = reads_x ? Xr : Yr;
            const int K = last ? 3 : 7, half = K >> 1;
            const int R0 = F - S - layer_lag(j);
            // conv_last with r <= 16 on one 16-channel tile (split-fp16 modes): shared by both body shapes
            auto conv_last16 = [&]() {
                // conv_last with r <= 16: one 16-channel output tile on v_mfma_f32_16x16x32_f16 instead of two
                // 32-channel tiles (of which 22+ channels are padding): every wave takes 48 rows of the step,
                // a quarter of the MFMA work.  D[out-ch][time]: lane (j = lane & 15, q4 = lane >> 4) ends up
                // with output channels 4 q4 .. 4 q4 + 3 of time row j.
                const int j16 = lane & 15, q4 = lane >> 4;
                const uint4* lw = reinterpret_cast<const uint4*>(p.last16) + lane;
                uint4 wh[BODY_CHUNKS_LAST], wl[BODY_CHUNKS_LAST];
#pragma unroll
                for (int cc = 0; cc < BODY_CHUNKS_LAST; ++cc) { wh[cc] = lw[(cc * 2) * 64]; wl[cc] = lw[(cc * 2 + 1) * 64]; }
#pragma unroll
                for (int f = 0; f < FRAGS_PER_CHUNK; ++f) { wf[0][f] = wload(0, f); wf[1][f] = wload(1, f); }   // next step
                const float4 b4 = ld4(biasl + 12 * 64 + 4 * q4);
                floatx4 a16[3];
                bool bad = false;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    a16[k][0] = b4.x; a16[k][1] = b4.y; a16[k][2] = b4.z; a16[k][3] = b4.w;
                    const int off = 48 * wave + 16 * k + j16;
#pragma unroll
                    for (int cc = 0; cc < BODY_CHUNKS_LAST; ++cc) {
                        const int d = cc >> 1, hh = cc & 1;
                        const char* row = src + ((R0 + off + d - 1) & (RING - 1)) * ROWB + (32 * hh + 8 * q4) * 2;
                        const half8 bh = as_h8(ldq(row)), bl = as_h8(ldq(row + 128));
                        a16[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wh[cc]), bh, a16[k], 0, 0, 0);
                        a16[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wh[cc]), bl, a16[k], 0, 0, 0);
                        a16[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(wl[cc]), bh, a16[k], 0, 0, 0);
                    }
                }
                STAMP_ADD(4);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int off = 48 * wave + 16 * k + j16;
                    const int g = R0 + off;
                    int nk, tk, nw, tw;
                    decode_row(nR, tR, off, nk, tk);
                    vmap(n0 + nk, tk, nw, tw);
                    const bool ok = (g >= 0) && (g < gend) && (tk < L) && (tw >= 0) && (tw < Ltrue) &&
                                    (tk >= p.halo) && (tk < p.halo + p.seg_len);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bad = bad || !(fabsf(a16[k][e]) <= 3.0e38f);
                    if (p.onset_ws != nullptr) {
                        // fused arg-max picker: per (virtual) waveform of this tile -- at most two, the tile is 16
                        // consecutive stream rows -- its max / min and the positions equal to the max
                        const bool lane_ok = ok && 4 * q4 < r;
                        float lm = -INFINITY, ll = INFINITY;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (lane_ok && 4 * q4 + e < r) { lm = fmaxf(lm, a16[k][e]); ll = fminf(ll, a16[k][e]); }
                        const int nv = n0 + nk;
                        const int nvA = __shfl(nv, 0), nvB = __shfl(nv, 15);
                        for (int pass = 0; pass < 2; ++pass) {
                            const int nvX = pass ? nvB : nvA;
                            if (pass && nvB == nvA) break;
                            const bool mine = lane_ok && nv == nvX;
                            const unsigned long long mm = __ballot(mine) & 0xffffull;      // rows of this waveform (q4 = 0 lanes)
                            if (mm == 0) continue;
                            float m = mine ? lm : -INFINITY, lo = mine ? ll : INFINITY;
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); lo = fminf(lo, __shfl_xor(lo, o)); }
                            const int jf = __builtin_ctzll(mm);
                            const int tw_base = __shfl(tw - j16, jf);
                            const int nwX = __shfl(nw, jf);
                            const int seg = nvX & seg_mask;
                            const int slot = seg * p.onset_seg_slots + (tw_base + jf - seg * p.seg_len + 15) / 16;
                            unsigned long long eq[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) eq[e] = __ballot(mine && 4 * q4 + e < r && a16[k][e] == m);
                            if (lane == 0) {
                                uint4* dst = reinterpret_cast<uint4*>(p.onset_ws + (size_t)nwX * p.onset_slots + slot);
                                dst[0] = make_uint4(1u, (unsigned)tw_base, __float_as_uint(m), __float_as_uint(lo));
                                dst[1] = make_uint4((unsigned)eq[0], (unsigned)(eq[0] >> 32), (unsigned)eq[1], (unsigned)(eq[1] >> 32));
                                dst[2] = make_uint4((unsigned)eq[2], (unsigned)(eq[2] >> 32), (unsigned)eq[3], (unsigned)(eq[3] >> 32));
                            }
                        }
                    }
                    if (!ok || 4 * q4 >= r || p.y == nullptr) continue;
                    float* const orow = p.y + ((size_t)nw * Ltrue + tw) * r + 4 * q4;
                    if ((r & 3) == 0) {
                        st4(orow, make_float4(a16[k][0], a16[k][1], a16[k][2], a16[k][3]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (4 * q4 + e < r) orow[e] = a16[k][e];
                    }
                }
                if (p.status != nullptr && __any(bad) && lane == 0) atomicOr(p.status, 1);
                STAMP_ADD(5);
                __syncthreads();
                STAMP_ADD(2);
            };
            if constexpr (SHAPE == 16) {
                // ---- split-fp16 layer on v_mfma_f32_16x16x32_f16 (stof_common.h "f16x3 body, 16x16x32").  Wave tile as
                // before (32 output channels x S/2 rows) = 2 M-tiles x NN N-tiles of 16 rows; one chunk = the whole K = 32
                // of an MFMA: 4 weight fragments (M-tile x hi | lo) and NN x 2 activation fragments feed 6 NN MFMAs.
                // Lane (i16 = time row of the N-tile, q4 = k-group / accumulator row group): accumulator element e of
                // acc[m][n] is output channel 32 mi + 8 q4 + 4 m + e of row R0 + 16 (NN ni + n) + i16.
                if (last && p.last16 != nullptr) {
                    conv_last16();
                    continue;
                }
                constexpr int NN = S / 32;
                const int i16 = lane & 15, q4 = lane >> 4;
                // the layer's bias enters as the C operand of each accumulator's first MFMA (no 48-register initialisation)
                floatx4 acc[2][NN], bvec[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float4 bb = ld4(biasl + j * 64 + 32 * mi + 8 * q4 + 4 * m);
                    bvec[m][0] = bb.x; bvec[m][1] = bb.y; bvec[m][2] = bb.z; bvec[m][3] = bb.w;
                }
                const int rbase = R0 + 16 * NN * ni + i16 - half;          // row of N-tile 0, tap 0
                const int cbyte = 16 * q4;                                  // k-group q4 = channels 8 q4 .. + 7 of the half
                auto bload = [&](uint4 (&b)[NN][2], int cc) {
                    const int d = cc >> 1, hh = cc & 1;
#pragma unroll
                    for (int n = 0; n < NN; ++n) {
                        const char* row = src + ((rbase + 16 * n + d) & (RING - 1)) * ROWB + 64 * hh + cbyte;
                        b[n][0] = ldq(row);
                        b[n][1] = ldq(row + 128);
                    }
                };
                auto mfma16 = [](const uint4& a, const uint4& b, floatx4 c) {
                    return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(a), as_h8(b), c, 0, 0, 0);
                };
                // hi*hi, hi*lo, lo*hi of one M-tile against all N-tiles, two N-tiles interleaved so that consecutive MFMAs
                // never share an accumulator
                auto mma_half = [&](const uint4& wh, const uint4& wl, const uint4 (&b)[NN][2], int m, bool first) {
#pragma unroll
                    for (int n = 0; n < NN; n += 2) {
                        acc[m][n] = mfma16(wh, b[n][0], first ? bvec[m] : acc[m][n]);
                        acc[m][n + 1] = mfma16(wh, b[n + 1][0], first ? bvec[m] : acc[m][n + 1]);
                        acc[m][n] = mfma16(wh, b[n][1], acc[m][n]);
                        acc[m][n + 1] = mfma16(wh, b[n + 1][1], acc[m][n + 1]);
                        acc[m][n] = mfma16(wl, b[n][0], acc[m][n]);
                        acc[m][n + 1] = mfma16(wl, b[n + 1][0], acc[m][n + 1]);
                    }
                };
                static_assert(NN % 2 == 0, "N-tiles are processed in pairs");
                // one chunk: M-tile 0 against every N-tile, refill its two weight fragments with those of chunk c+2 at once
                // (their last use is behind them; the loads then have a whole chunk to land), the same for M-tile 1; the
                // ds_reads of the NEXT chunk's activation fragments ride behind every third MFMA.  The VMEM groups pin the
                // refills where they are written: left to itself the scheduler sinks them to just before their use.
                auto do_chunk = [&](uint4 (&w)[FRAGS_PER_CHUNK], uint4 (&bcur)[NN][2], uint4 (&bnext)[NN][2], int cc) {
                    const int c2 = (c + 2 >= NCHUNK_STEP) ? c + 2 - NCHUNK_STEP : c + 2;
                    bload(bnext, cc + 1);          // past the layer's last chunk this reads rows nobody uses
                    mma_half(w[0], w[1], bcur, 0, cc == 0);
                    w[0] = wload(c2, 0);
                    w[1] = wload(c2, 1);
                    mma_half(w[2], w[3], bcur, 1, cc == 0);
                    w[2] = wload(c2, 2);
                    w[3] = wload(c2, 3);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int i = 0; i < NN; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                    }
                    ++c;
                };
                uint4 bf0[NN][2], bf1[NN][2];
                bload(bf0, 0);
                STAMP_ADD(3);                         // layer setup + first activation fragments
                char* const dst = (j & 1) ? Yr : Xr;                // odd sweep layers write ring Y
                // forward: residual add for conv3,5,..,11 and conv12, leaky ReLU for conv2,4,..,10; backward: j = 1 plain, even j
                // times lrelu'(saved activation), odd j >= 3 added in place to the residual gradient
                const bool inplace = BWD ? ((j & 1) && j >= 3) : (!(j & 1) || (j == 11));
                // validity and LDS slot of the wave's row of N-tile n
                auto row_of = [&](int n, bool& valid, int& slot, int& nw, int& tw, int& tk) {
                    const int off = 16 * (NN * ni + n) + i16;
                    const int g = R0 + off;
                    int nk;
                    decode_row(nR, tR, off, nk, tk);
                    vmap(n0 + nk, tk, nw, tw);
                    valid = (g >= 0) && (g < gend) && (tk < L) && (tw >= 0) && (tw < Ltrue);
                    slot = (g & (RING - 1)) * ROWB + (32 * mi + 8 * q4) * 2;
                };
                if (!last) {
                    // ---- main part chunk-major, then the last TAILC chunks tile-major: the epilogue of N-tile n-1 (residual add
                    // or leaky ReLU, validity, fp16 split, ring store: VALU + LDS work) is issued in the shadow of the MFMAs of
                    // N-tile n; only the last tile's epilogue stays exposed (1/6 of the layer's output, 1/3 with 32-row tiles).
                    // Wave-uniform test on the wave's S/2-row span: inside one waveform (and, in segment mode, inside the segment's
                    // own rows), nothing but real samples?  Then no row needs masking and the training dump addresses are affine.
                    const int offw0 = 16 * NN * ni, gw0 = R0 + offw0;
                    int nkw, tkw, nww, tww;
                    decode_row(nR, tR, offw0, nkw, tkw);
                    vmap(n0 + nkw, tkw, nww, tww);
                    const bool span_ok = (gw0 >= 0) && (gw0 + 16 * NN - 1 < gend) && (tkw + 16 * NN - 1 < L) && (tww >= 0) &&
                                         (tww + 16 * NN - 1 < Ltrue) && (tkw >= p.halo) && (tkw + 16 * NN - 1 < p.halo + p.seg_len);
                    // training dump of this layer's output (tensor j): lane = 8 consecutive channels of its row of every N-tile;
                    // a span with padding rows writes to the scratch tail instead and is dumped row by row afterwards
                    float* const dumpj = DUMP ? p.dump + (size_t)j * p.dump_stride : nullptr;
                    float* dlane = nullptr;
                    long long dstep = 0;
                    if constexpr (DUMP) {
                        if (span_ok) {
                            dlane = dumpj + ((size_t)nww * Ltrue + tww + i16) * NF + 32 * mi + 8 * q4;
                            dstep = 16 * NF;
                        } else {
                            dlane = p.dump + 12 * p.dump_stride + 8 * lane;
                        }
                    }
                    // backward, masked layers: the saved activation ys[k] (the forward dump's tensor 1 + 2 k) of the lane's 8 channels
                    // of its row of every N-tile -- requested here, before the main chunks, so the loads have a whole layer's MFMAs
                    // to arrive (the backward kernel has the registers: no x0 pass, no conv_last)
                    const bool masked = BWD && !(j & 1);                       // conv(2k+3)^T output times lrelu'(ys[k]), k = 5 - j/2
                    float4 ysv[BWD ? NN : 1][2];
                    if constexpr (BWD) {
#pragma unroll
                        for (int n = 0; n < NN; ++n) ysv[n][0] = ysv[n][1] = make_float4(1.f, 1.f, 1.f, 1.f);
                        if (masked) {
                            const float* const ys = p.fwd_dump + (size_t)(1 + 2 * (5 - (j >> 1))) * p.dump_stride + 32 * mi + 8 * q4;
                            if (span_ok) {
                                const float* const src0 = ys + ((size_t)nww * Ltrue + tww + i16) * NF;
#pragma unroll
                                for (int n = 0; n < NN; ++n) { ysv[n][0] = ld4(src0 + (size_t)n * 16 * NF); ysv[n][1] = ld4(src0 + (size_t)n * 16 * NF + 4); }
                            } else {
#pragma unroll
                                for (int n = 0; n < NN; ++n) {
                                    bool valid;
                                    int slot, nw, tw, tk;
                                    row_of(n, valid, slot, nw, tw, tk);
                                    if (valid) {
                                        ysv[n][0] = ld4(ys + ((size_t)nw * Ltrue + tw) * NF);
                                        ysv[n][1] = ld4(ys + ((size_t)nw * Ltrue + tw) * NF + 4);
                                    }
                                }
                            }
                        }
                    }
#ifndef STOF_TAILC
#define STOF_TAILC 2
#endif
                    constexpr int TAILC = STOF_TAILC, MAINC = BODY_CHUNKS_K7 - TAILC;       // A/B: 0 = no tile-major tail, whole epilogue exposed
#pragma unroll
                    for (int cc = 0; cc < MAINC; cc += 2) {
                        do_chunk(wf[0], bf0, bf1, cc);
                        do_chunk(wf[1], bf1, bf0, cc + 1);
                    }
                    // here: wf[0] / wf[1] = the layer's last two chunks, bf0 = activation fragments of chunk MAINC (all tiles);
                    // bf1 is free: it takes the fragments of the last chunk, which land while tile 0's first MFMAs run
                    if (TAILC > 0) bload(bf1, MAINC + 1);
#ifdef STOF_STAMP_TAIL_AS_EPI
                    STAMP_ADD(4);                     // diagnostic: the tile-major tail is then counted with the epilogue (slot 5)
#endif
                    // MFMA k (0..11) of N-tile n in the tail: hi*hi, hi*lo, lo*hi of both M-tiles, interleaved so that consecutive
                    // MFMAs never share an accumulator; k < 6 on the second-to-last chunk, k >= 6 on the last one
                    auto tail_mfma = [&](int n, int k) {
                        const uint4 (&w)[FRAGS_PER_CHUNK] = k < 6 ? wf[0] : wf[1];
                        const uint4 (&b)[2] = k < 6 ? bf0[n] : bf1[n];
                        const int kk = k % 6, m = kk & 1;
                        acc[m][n] = mfma16(w[2 * m + (kk >= 4 ? 1 : 0)], b[(kk >> 1) == 1 ? 1 : 0], acc[m][n]);
                    };
                    const float one = opaque_one();
                    // The tail exists twice (in-place layers: old value + add; activation layers: leaky ReLU) so that the slices
                    // carry only the VALU work their layer needs.
                    // kinds: 0 = residual add in place, 1 = leaky ReLU (forward), 2 = plain, 3 = times lrelu'(saved) (backward)
                    auto tail = [&](auto kind_c) {
                        constexpr int KIND = decltype(kind_c)::value;
                        constexpr bool INPL = KIND == 0;
                        struct Epi { int slot; uint4 oh, ol; float v[4]; half2v h01, h23; unsigned hi[4], lo[4]; };
                        // The epilogue of one N-tile (8 consecutive channels of one row per lane) in 12 slices of <= 4 VALU
                        // instructions, one behind each MFMA of the NEXT tile:  0: LDS slot + (in-place) loads of the old hi | lo
                        // images -- issued one tile early, nothing waits for LDS inside a region;  1-5 / 6-10: M-tile 0 / 1:
                        // accumulator read, residual add (two mixed-precision FMAs per value: + hi, + lo) or leaky ReLU, fp16 split;
                        // 11: the two 16-byte stores.  Branch-free; rows that are padding are zeroed afterwards (rare).
                        auto epi_slice = [&](Epi& e, int n, int k) {
                            if (k == 0) {
                                e.slot = ((R0 + 16 * (NN * ni + n) + i16) & (RING - 1)) * ROWB + (32 * mi + 8 * q4) * 2;
                                if constexpr (INPL) { e.oh = ldq(dst + e.slot); e.ol = ldq(dst + e.slot + 128); }
                                return;
                            }
                            if (k == 11) {
                                *reinterpret_cast<uint4*>(dst + e.slot) = make_uint4(e.hi[0], e.hi[1], e.hi[2], e.hi[3]);
                                *reinterpret_cast<uint4*>(dst + e.slot + 128) = make_uint4(e.lo[0], e.lo[1], e.lo[2], e.lo[3]);
                                return;
                            }
                            const int m = (k - 1) / 5, st = (k - 1) % 5;
                            if (st == 0) {
#pragma unroll
                                for (int x = 0; x < 4; ++x) e.v[x] = acc[m][n][x];
                            } else if (st <= 2) {
                                const int x0 = 2 * (st - 1);                  // values x0, x0 + 1 = one packed pair of the old images
                                if constexpr (INPL) {
                                    const unsigned hw = m ? (st == 1 ? e.oh.z : e.oh.w) : (st == 1 ? e.oh.x : e.oh.y);
                                    const unsigned lw = m ? (st == 1 ? e.ol.z : e.ol.w) : (st == 1 ? e.ol.x : e.ol.y);
                                    const half2v h = bits_h2(hw), l = bits_h2(lw);
                                    e.v[x0] = mix_add(l[0], mix_add(h[0], e.v[x0], one), one);
                                    e.v[x0 + 1] = mix_add(l[1], mix_add(h[1], e.v[x0 + 1], one), one);
                                } else if constexpr (KIND == 1) {
                                    // leaky_relu(v, 0.01) = max(v, 0.01 v) = med3(v, 0.01 v, huge): one op, where fmaxf costs a canonicalising v_max on top
                                    e.v[x0] = __builtin_amdgcn_fmed3f(e.v[x0], 0.01f * e.v[x0], 3.0e38f);
                                    e.v[x0 + 1] = __builtin_amdgcn_fmed3f(e.v[x0 + 1], 0.01f * e.v[x0 + 1], 3.0e38f);
                                } else if constexpr (KIND == 3) {
                                    // times lrelu'(saved activation): 1 where it was positive, 0.01 elsewhere
                                    const float4 sv = ysv[n][m];
                                    const float s0 = x0 == 0 ? sv.x : sv.z, s1 = x0 == 0 ? sv.y : sv.w;
                                    e.v[x0] = s0 > 0.f ? e.v[x0] : 0.01f * e.v[x0];
                                    e.v[x0 + 1] = s1 > 0.f ? e.v[x0 + 1] : 0.01f * e.v[x0 + 1];
                                }
                            } else if (st == 3) {
                                if constexpr (DUMP) st4(dlane + n * dstep + 4 * m, make_float4(e.v[0], e.v[1], e.v[2], e.v[3]));
                                e.h01 = cvt_h2(e.v[0], e.v[1]);
                                e.h23 = cvt_h2(e.v[2], e.v[3]);
                                e.v[0] = mix_sub(e.h01[0], e.v[0], one);
                                e.v[1] = mix_sub(e.h01[1], e.v[1], one);
                            } else {
                                e.v[2] = mix_sub(e.h23[0], e.v[2], one);
                                e.v[3] = mix_sub(e.h23[1], e.v[3], one);
                                e.hi[2 * m] = h2_bits(e.h01); e.hi[2 * m + 1] = h2_bits(e.h23);
                                e.lo[2 * m] = h2_bits(cvt_h2(e.v[0], e.v[1]));
                                e.lo[2 * m + 1] = h2_bits(cvt_h2(e.v[2], e.v[3]));
                            }
                        };
                        // tile-major over the last two chunks: MFMA k of N-tile n, then slice k of N-tile n-1's epilogue (slice 0: of
                        // its own) in program order; one scheduling region per tile with (1 MFMA, 4 VALU) groups.  (Given the
                        // epilogue as four coarse pieces the group-barrier solver clumped the VALU work between two MFMAs.)
                        Epi ep[2];
                        if constexpr (TAILC == 0) {
#pragma unroll
                            for (int n = 0; n < NN; ++n) {
#pragma unroll
                                for (int k = 0; k < 12; ++k) epi_slice(ep[0], n, k);
                            }
                            return;
                        }
#pragma unroll
                        for (int n = 0; n < NN; ++n) {
#pragma unroll
                            for (int k = 0; k < 12; ++k) {
                                tail_mfma(n, k);
                                if (k == 0) epi_slice(ep[n & 1], n, 0);
                                else if (n > 0) epi_slice(ep[(n - 1) & 1], n - 1, k);
                                if (n == NN - 1 && (k == 5 || k == 11)) {   // last use of wf[0] / wf[1]: refill with the next layer's first chunks
                                    const int q = k == 5 ? 0 : 1;
                                    const int c2 = (c + 2 + q >= NCHUNK_STEP) ? c + 2 + q - NCHUNK_STEP : c + 2 + q;
#pragma unroll
                                    for (int f = 0; f < FRAGS_PER_CHUNK; ++f) wf[q][f] = wload(c2, f);
                                }
#ifdef STOF_TAIL_BLOCK_SCHED
                                __builtin_amdgcn_sched_barrier(0);      // A/B: pin every (MFMA, slice) pair; measured 0.5 % slower
#endif
                            }
#ifndef STOF_TAIL_BLOCK_SCHED
#pragma unroll
                            for (int i = 0; i < 12; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#endif
                        }
                        c += TAILC;
#ifdef STOF_STAMP_TAIL_AS_EPI
                        STAMP_ADD(5);
#else
                        STAMP_ADD(4);                     // chunk loop (MFMA) incl. the overlapped epilogues
#endif
#pragma unroll
                        for (int k = 1; k < 12; ++k) epi_slice(ep[(NN - 1) & 1], NN - 1, k);
                    };
                    if (inplace) tail(std::integral_constant<int, 0>{});
                    else if constexpr (BWD) {
                        if (j == 1) tail(std::integral_constant<int, 2>{});
                        else tail(std::integral_constant<int, 3>{});
                    } else tail(std::integral_constant<int, 1>{});
                    // Rows outside [0, L) of their waveform (gap rows, stream ends, segment padding) must read as zeros for the
                    // next layer (= its zero padding).  When the wave's span lies inside one waveform nothing is to do (the
                    // common case); otherwise the lanes of padding rows overwrite what the branch-free tail stored (same lane,
                    // program order: no race; the barrier follows), and the training dump takes the valid rows of the span
                    // from the LDS image (hi + lo = the value the next layer reads).
                    if (!span_ok) {
#pragma unroll 1
                        for (int n = 0; n < NN; ++n) {
                            bool valid;
                            int slot, nw, tw, tk;
                            row_of(n, valid, slot, nw, tw, tk);
                            if (!valid) {
                                *reinterpret_cast<uint4*>(dst + slot) = make_uint4(0u, 0u, 0u, 0u);
                                *reinterpret_cast<uint4*>(dst + slot + 128) = make_uint4(0u, 0u, 0u, 0u);
                            } else if (DUMP && tk >= p.halo && tk < p.halo + p.seg_len) {
                                const half8 hh8 = as_h8(ldq(dst + slot)), ll8 = as_h8(ldq(dst + slot + 128));
                                float* const o = dumpj + ((size_t)nw * Ltrue + tw) * NF + 32 * mi + 8 * q4;
                                st4(o, make_float4((float)hh8[0] + (float)ll8[0], (float)hh8[1] + (float)ll8[1],
                                                   (float)hh8[2] + (float)ll8[2], (float)hh8[3] + (float)ll8[3]));
                                st4(o + 4, make_float4((float)hh8[4] + (float)ll8[4], (float)hh8[5] + (float)ll8[5],
                                                       (float)hh8[6] + (float)ll8[6], (float)hh8[7] + (float)ll8[7]));
                            }
                        }
                    }
                    STAMP_ADD(5);                     // exposed epilogue (last tile)
                    __syncthreads();
                    STAMP_ADD(2);
                    continue;
                }
                // ---- conv_last with r > 16 (the 16-channel tile of conv_last16 serves r <= 16): 6 chunks, outputs to HBM
#pragma unroll
                for (int cc = 0; cc < BODY_CHUNKS_LAST; cc += 2) { do_chunk(wf[0], bf0, bf1, cc); do_chunk(wf[1], bf1, bf0, cc + 1); }
                STAMP_ADD(4);                         // chunk loop (MFMA)
                bool bad = false;
#pragma unroll
                for (int n = 0; n < NN; ++n) {
                    bool valid;
                    int slot, nw, tw, tk;
                    row_of(n, valid, slot, nw, tw, tk);
                    // conv_last + SampleShuffle1D: out[n][t*r + k] = conv_last[n][k][t]; only the segment's own rows
                    valid = valid && (tk >= p.halo) && (tk < p.halo + p.seg_len);
                    const float v[8] = {acc[0][n][0], acc[0][n][1], acc[0][n][2], acc[0][n][3],
                                        acc[1][n][0], acc[1][n][1], acc[1][n][2], acc[1][n][3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) bad = bad || !(fabsf(v[e]) <= 3.0e38f);
                    if (!valid || p.y == nullptr) continue;
                    float* const orow = p.y + ((size_t)nw * Ltrue + tw) * r;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int c0 = 32 * mi + 8 * q4 + 4 * m;
                        if (c0 >= r) continue;
                        if ((r & 3) == 0) {
                            st4(orow + c0, make_float4(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c0 + e < r) orow[c0 + e] = v[4 * m + e];
                        }
                    }
                }
                if (p.status != nullptr && __any(bad) && lane == 0) atomicOr(p.status, 1);
                STAMP_ADD(5);                         // epilogue
                __syncthreads();
                STAMP_ADD(2);
            } else {
            // (for conv_last with r <= 32 the waves of the upper output tile multiply zero-padded
            //  weights: free in wall time, and it keeps the chunk body branch-free)
            // accumulators start at the layer's bias (lane (ln, lh) holds channels 32mi + 8gg + 4lh + e)
            floatx16 acc[NT];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const float4 bb = ld4(biasl + j * 64 + 32 * mi + 8 * gg + 4 * lh);
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    acc[k][4 * gg] = bb.x; acc[k][4 * gg + 1] = bb.y; acc[k][4 * gg + 2] = bb.z; acc[k][4 * gg + 3] = bb.w;
                }
            }

            // activation fragments of chunk cc: NT row tiles x 4 fragments (ds_read_b128 each)
            auto bload = [&](uint4 (&b)[NT][FRAGS_PER_CHUNK], int cc) {
                const int d = cc >> 1, hh = cc & 1;
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int g = R0 + 32 * (NT * ni + k) + ln + d - half;
                    const char* row = src + (g & (RING - 1)) * ROWB;
#pragma unroll
                    for (int f = 0; f < FRAGS_PER_CHUNK; ++f) b[k][f] = ldq(row + act_frag_off<PREC>(f, hh, lh));
                }
            };
            // one chunk: MFMAs on (w, bcur) while the ds_reads of the next chunk (bnext) are in flight,
            // then refill w with the fragments of chunk c+2
            auto do_chunk = [&](uint4 (&w)[FRAGS_PER_CHUNK], uint4 (&bcur)[NT][FRAGS_PER_CHUNK],
                                uint4 (&bnext)[NT][FRAGS_PER_CHUNK], int cc) {
                const int c2 = (c + 2 >= BODY_NCHUNK) ? c + 2 - BODY_NCHUNK : c + 2;
                bload(bnext, cc + 1);          // past the layer's last chunk this reads rows nobody uses
                if constexpr (PREC == STOF_PREC_FP32) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int k = 0; k < NT; ++k) acc[k] = mma_fp32(w[q], bcur[k][q], acc[k]);
                        w[q] = wload(c2, q);
                    }
                    // interleave: one ds_read behind each of the first MFMAs
#pragma unroll
                    for (int i = 0; i < NT * FRAGS_PER_CHUNK; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                        for (int k = 0; k < NT; ++k)
                            acc[k] = mma_f16x3(w[2 * ks], w[2 * ks + 1], bcur[k][2 * ks], bcur[k][2 * ks + 1], acc[k]);
                        w[2 * ks] = wload(c2, 2 * ks);
                        w[2 * ks + 1] = wload(c2, 2 * ks + 1);
                    }
#pragma unroll
                    for (int i = 0; i < NT * FRAGS_PER_CHUNK; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                ++c;
            };
            // The chunk loop is fully unrolled (14 chunks for k7 layers, 6 for conv_last): in straight-line
            // code the compiler counts s_waitcnt vmcnt(N) for the weight prefetch instead of draining it
            // at a loop header, and the two register sets alternate statically.
            uint4 bf0[NT][FRAGS_PER_CHUNK], bf1[NT][FRAGS_PER_CHUNK];
            bload(bf0, 0);
            auto run_chunks = [&](auto nchunk_c) {
                constexpr int NCH = decltype(nchunk_c)::value;
#pragma unroll
                for (int cc = 0; cc < NCH; cc += 2) {
                    do_chunk(wf[0], bf0, bf1, cc);
                    do_chunk(wf[1], bf1, bf0, cc + 1);
                }
            };
            STAMP_ADD(3);                         // layer setup + first activation fragments
#ifndef STOF_NO_TAIL
            // Tail overlap (k7 layers): the last TAILC chunks run tile-major, so the epilogue of row tile k
            // (residual add / leaky ReLU, fp16 split, ring store -- VALU + LDS work) is issued in the shadow of
            // the MFMAs of tile k+1; only the last tile's epilogue stays exposed.  Program order already
            // interleaves the LDS accesses (the compiler cannot reorder ds ops it cannot disambiguate); the
            // sched_group_barriers only slide the VALU work between the MFMAs.
            constexpr int TAILC = 4;
            bool tvalid[NT];
            int tslot[NT];
            if (!last) {
                char* const dst = (j & 1) ? Yr : Xr;                // odd sweep layers write ring Y
                const bool inplace = !(j & 1) || (j == 11);         // residual add: conv3,5,..,11 and conv12
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int off = 32 * (NT * ni + k) + ln;
                    const int g = R0 + off;
                    int nk, tk;
                    decode_row(nR, tR, off, nk, tk);
                    int nw, tw;
                    vmap(n0 + nk, tk, nw, tw);
                    tvalid[k] = (g >= 0) && (g < gend) && (tk < L) && (tw >= 0) && (tw < Ltrue);
                    tslot[k] = (g & (RING - 1)) * ROWB + (32 * mi + 4 * lh) * (PREC == STOF_PREC_FP32 ? 4 : 2);
                }
                constexpr int MAINC = BODY_CHUNKS_K7 - TAILC;
                uint4 wC[FRAGS_PER_CHUNK], wD[FRAGS_PER_CHUNK];
#pragma unroll
                for (int cc = 0; cc < MAINC; cc += 2) {
                    if (cc == MAINC - 2) {                          // weights of the last two chunks of the layer
#pragma unroll
                        for (int f = 0; f < FRAGS_PER_CHUNK; ++f) { wC[f] = wload(c + 4, f); wD[f] = wload(c + 5, f); }
                    }
                    do_chunk(wf[0], bf0, bf1, cc);
                    do_chunk(wf[1], bf1, bf0, cc + 1);
                }
                // here: wf[0] / wf[1] / wC / wD = chunks MAINC .. MAINC+3, bf0 = fragments of chunk MAINC (all tiles)
                auto bload1 = [&](uint4 (&b)[FRAGS_PER_CHUNK], int k, int cc) {
                    const int d = cc >> 1, hh = cc & 1;
                    const int g = R0 + 32 * (NT * ni + k) + ln + d - half;
                    const char* row = src + (g & (RING - 1)) * ROWB;
#pragma unroll
                    for (int f = 0; f < FRAGS_PER_CHUNK; ++f) b[f] = ldq(row + act_frag_off<PREC>(f, hh, lh));
                };
                auto mma_tile = [&](const uint4 (&w)[FRAGS_PER_CHUNK], const uint4 (&b)[FRAGS_PER_CHUNK], floatx16& a) {
                    if constexpr (PREC == STOF_PREC_FP32) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) a = mma_fp32(w[q], b[q], a);
                    } else {
                        a = mma_f16x3(w[0], w[1], b[0], b[1], a);
                        a = mma_f16x3(w[2], w[3], b[2], b[3], a);
                    }
                };
                // The tail exists twice, specialised for the two kinds of k7 layers, so that the epilogue pieces carry only
                // the VALU work their layer needs (in-place layers: old value + add; activation layers: leaky ReLU):
                // that is what makes a piece fit into the shadow of one chunk's MFMAs.
                const float validf[NT] = {tvalid[0] ? 1.f : 0.f, tvalid[NT > 1 ? 1 : 0] ? 1.f : 0.f, tvalid[NT - 1] ? 1.f : 0.f};
                static_assert(NT == 3, "validf initialiser assumes 3 row tiles per wave");
                auto tail = [&](auto inplace_c) {
                    constexpr bool INPL = decltype(inplace_c)::value;
                    // `old` is the raw LDS image of the 4 channels (fp32: float4; f16x3: hi | lo halves)
                    auto epi_load = [&](int k, int gg) -> uint4 {
                        if constexpr (!INPL) return make_uint4(0u, 0u, 0u, 0u);
                        const char* q = dst + tslot[k];
                        if constexpr (PREC == STOF_PREC_FP32) {
                            return ldq(q + 32 * gg);
                        } else {
                            const uint2 h = *reinterpret_cast<const uint2*>(q + 16 * gg);
                            const uint2 l = *reinterpret_cast<const uint2*>(q + 128 + 16 * gg);
                            return make_uint4(h.x, h.y, l.x, l.y);
                        }
                    };
                    auto epi_store = [&](int k, int gg, uint4 old) {
                        float2v v01 = {acc[k][4 * gg], acc[k][4 * gg + 1]}, v23 = {acc[k][4 * gg + 2], acc[k][4 * gg + 3]};
                        char* q = dst + tslot[k];
                        if constexpr (INPL) {
                            if constexpr (PREC == STOF_PREC_FP32) {
                                const float4 o = as_f4(old);
                                v01[0] += o.x; v01[1] += o.y; v23[0] += o.z; v23[1] += o.w;
                            } else {
                                union { uint4 u; half2v h[4]; } c;
                                c.u = old;
                                v01 += __builtin_convertvector(c.h[0], float2v) + __builtin_convertvector(c.h[2], float2v);
                                v23 += __builtin_convertvector(c.h[1], float2v) + __builtin_convertvector(c.h[3], float2v);
                            }
                        }
                        const float2v vf = {validf[k], validf[k]};
                        v01 *= vf; v23 *= vf;
                        if constexpr (!INPL) {
                            const float2v sl = {0.01f, 0.01f};
                            const float2v t01 = v01 * sl, t23 = v23 * sl;
                            v01[0] = fmaxf(v01[0], t01[0]); v01[1] = fmaxf(v01[1], t01[1]);
                            v23[0] = fmaxf(v23[0], t23[0]); v23[1] = fmaxf(v23[1], t23[1]);
                        }
                        if constexpr (PREC == STOF_PREC_FP32) {
                            *reinterpret_cast<float4*>(q + 32 * gg) = make_float4(v01[0], v01[1], v23[0], v23[1]);
                        } else {
                            const half2v h01 = __builtin_convertvector(v01, half2v), h23 = __builtin_convertvector(v23, half2v);
                            const float2v d01 = v01 - __builtin_convertvector(h01, float2v);
                            const float2v d23 = v23 - __builtin_convertvector(h23, float2v);
                            const half2v l01 = __builtin_convertvector(d01, half2v), l23 = __builtin_convertvector(d23, half2v);
                            union { half2v h[2]; uint2 u; } ph, pl;
                            ph.h[0] = h01; ph.h[1] = h23; pl.h[0] = l01; pl.h[1] = l23;
                            *reinterpret_cast<uint2*>(q + 16 * gg) = ph.u;
                            *reinterpret_cast<uint2*>(q + 128 + 16 * gg) = pl.u;
                        }
                    };
                    uint4 bq[2][FRAGS_PER_CHUNK];
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
#pragma unroll
                        for (int q = 0; q < TAILC; ++q) {
                            uint4 o = make_uint4(0u, 0u, 0u, 0u);
                            if (k > 0) o = epi_load(k - 1, q);
                            if (q + 1 < TAILC) bload1(bq[(q + 1) & 1], k, MAINC + q + 1);
                            uint4 (&wq)[FRAGS_PER_CHUNK] = q == 0 ? wf[0] : q == 1 ? wf[1] : q == 2 ? wC : wD;
                            uint4 (&bcur)[FRAGS_PER_CHUNK] = q == 0 ? bf0[k] : bq[q & 1];
                            mma_tile(wq, bcur, acc[k]);
                            if (k == NT - 1 && q < 2) {             // last use of wf[q]: refill with the next layer's chunk q
                                const int c2 = (c + 4 + q >= BODY_NCHUNK) ? c + 4 + q - BODY_NCHUNK : c + 4 + q;
#pragma unroll
                                for (int f = 0; f < FRAGS_PER_CHUNK; ++f) wf[q][f] = wload(c2, f);
                            }
                            if (k > 0) epi_store(k - 1, q, o);
                            constexpr int NM = PREC == STOF_PREC_FP32 ? 16 : 6;
#pragma unroll
                            for (int i = 0; i < NM; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, PREC == STOF_PREC_FP32 ? 2 : 5, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);      // one scheduling region per (tile, chunk)
                        }
                    }
                    c += TAILC;
                    STAMP_ADD(4);                     // chunk loop (MFMA) incl. the overlapped epilogues
                    uint4 o[4];
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) o[gg] = epi_load(NT - 1, gg);
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) epi_store(NT - 1, gg, o[gg]);
                };
                if (inplace) tail(std::true_type{});
                else tail(std::false_type{});
                STAMP_ADD(5);                     // exposed epilogue (last tile)
                __syncthreads();
                STAMP_ADD(2);
                continue;
            }
            if constexpr (PREC == STOF_PREC_F16X3) {
                if (p.last16 != nullptr) {
                    conv_last16();
                    continue;
                }
            }
            run_chunks(std::integral_constant<int, BODY_CHUNKS_LAST>{});
#else
            if (last) run_chunks(std::integral_constant<int, BODY_CHUNKS_LAST>{});
            else run_chunks(std::integral_constant<int, BODY_CHUNKS_K7>{});
#endif
            STAMP_ADD(4);                         // chunk loop (MFMA)

            // ---- epilogue of sweep layer j (the destination ring is not read by this layer)
            {
                bool valid[NT];
                int slot[NT], tt[NT], nn[NT];
                bool allvalid = true;
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int off = 32 * (NT * ni + k) + ln;
                    const int g = R0 + off;
                    decode_row(nR, tR, off, nn[k], tt[k]);
                    int nw, tw;
                    vmap(n0 + nn[k], tt[k], nw, tw);
                    valid[k] = (g >= 0) && (g < gend) && (tt[k] < L) && (tw >= 0) && (tw < Ltrue);
                    if (last) {                                      // outputs: only the segment's own rows
                        valid[k] = valid[k] && (tt[k] >= p.halo) && (tt[k] < p.halo + p.seg_len);
                        nn[k] = nw - n0;
                        tt[k] = tw;
                    }
                    slot[k] = g & (RING - 1);
                    allvalid = allvalid && valid[k];
                }
                const bool wave_all_valid = __all(allvalid);          // wave-uniform fast path: no masking
                if (!last) {
                    char* const dst = (j & 1) ? Yr : Xr;            // odd sweep layers write ring Y
                    const bool inplace = !(j & 1) || (j == 11);     // residual add: conv3,5,..,11 and conv12
                    const bool act = (j & 1) && (j != 11);          // leaky ReLU: conv2,4,..,10
                    float4 v[NT][4];
#pragma unroll
                    for (int k = 0; k < NT; ++k)
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg)
                            v[k][gg] = make_float4(acc[k][4 * gg], acc[k][4 * gg + 1], acc[k][4 * gg + 2], acc[k][4 * gg + 3]);
                    if (inplace) {
                        // all residual reads first (one latency), then the adds
                        float4 o[NT][4];
#pragma unroll
                        for (int k = 0; k < NT; ++k)
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg)
                                o[k][gg] = load_act4<PREC>(dst + slot[k] * ROWB, 32 * mi + 8 * gg + 4 * lh);
#pragma unroll
                        for (int k = 0; k < NT; ++k)
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) {
                                v[k][gg].x += o[k][gg].x; v[k][gg].y += o[k][gg].y;
                                v[k][gg].z += o[k][gg].z; v[k][gg].w += o[k][gg].w;
                            }
                    } else if (act) {
#pragma unroll
                        for (int k = 0; k < NT; ++k)
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) {
                                // leaky_relu(v, 0.01) = max(v, 0.01 v)
                                v[k][gg].x = fmaxf(v[k][gg].x, 0.01f * v[k][gg].x);
                                v[k][gg].y = fmaxf(v[k][gg].y, 0.01f * v[k][gg].y);
                                v[k][gg].z = fmaxf(v[k][gg].z, 0.01f * v[k][gg].z);
                                v[k][gg].w = fmaxf(v[k][gg].w, 0.01f * v[k][gg].w);
                            }
                    }
                    if (!wave_all_valid) {
#pragma unroll
                        for (int k = 0; k < NT; ++k)
                            if (!valid[k]) {
#pragma unroll
                                for (int gg = 0; gg < 4; ++gg) v[k][gg] = make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                    }
#pragma unroll
                    for (int k = 0; k < NT; ++k)
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg)
                            store_act4<PREC>(dst + slot[k] * ROWB, 32 * mi + 8 * gg + 4 * lh, v[k][gg]);
                } else {
                    // conv_last + SampleShuffle1D: out[n][t*r + k] = conv_last[n][k][t]
                    if (p.status != nullptr) {
                        // An activation beyond the fp16 range (f16x3 mode) turns into inf/NaN and stays so in
                        // every output of its receptive field, so checking the last layer alone is enough.
                        bool bad = false;
#pragma unroll
                        for (int k = 0; k < NT; ++k)
#pragma unroll
                            for (int e = 0; e < 16; ++e) bad = bad || !(fabsf(acc[k][e]) <= 3.0e38f);
                        if (__any(bad) && lane == 0) atomicOr(p.status, 1);
                    }
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        if (!valid[k]) continue;
                        float* const orow = p.y + ((size_t)(n0 + nn[k]) * Ltrue + tt[k]) * r;
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) {
                            const int c0 = 32 * mi + 8 * gg + 4 * lh;
                            if (c0 >= r) continue;
                            const float vv[4] = {acc[k][4 * gg], acc[k][4 * gg + 1], acc[k][4 * gg + 2], acc[k][4 * gg + 3]};
                            if ((r & 3) == 0) {
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
            STAMP_ADD(5);                         // epilogue
            __syncthreads();
            STAMP_ADD(2);
            }   // SHAPE == 32
        }
    }
#ifdef STOF_STAMPS
    if (lane == 0 && p.stamps) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = tsum[i];
        o[6] = stamp() - tstart;
        o[7] = (unsigned long long)nsteps;
    }
#endif
}

constexpr int BODY_P2_C1F = 640;        // conv1 taps + bias [64][10] kept in LDS by the two-pass sweep (re-read per x0 pass: frees 40 registers)
#include "body_p2.h"      // round 4: the split-fp16 sweep with two-pass tile-major k7 layers (the default)

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
    const int* run_if;     // as BodyParams::run_if
    unsigned char* arg;    // training (ARG): [N][P][512] row offset (0..79) of the window's FIRST maximum, for the pool's backward
};

template <int PREC, int NW, int SHAPE = 32, bool ARG = false>
__global__ __launch_bounds__(256, SGB_WAVES_PER_SIMD) void sgb_contract_pool_kernel(const SgbParams p) {
    static_assert(!ARG || SHAPE == 16, "the arg-max output lives in the 16x16x32 form");
    static_assert(NW % 2 == 0, "80*NW must be a multiple of 32");
    static_assert(SHAPE == 32 || (SHAPE == 16 && PREC == STOF_PREC_F16X3), "the 16x16x32 form is a split-fp16 kernel");
    constexpr int ROWS = SGB_SCALE * NW;          // output rows of the tile
    constexpr int MT = ROWS / 32;
    constexpr int TR = ROWS + 4;                  // conv1 rows needed (k5: +-2)
    constexpr int RAWN = TR + 8;                  // raw samples needed (k9: +-4)
    constexpr int RF = SHAPE == 16 ? ROWF16 : ROWF;           // 288-byte rows for the 16x16x32 operand reads (see BodyLds)
    constexpr int ROWB = RF * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const act = reinterpret_cast<char*>(smem);          // [TR] rows of ROWB bytes
    float* const raw = smem + TR * RF;                         // [RAWN]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;
    const int n = blockIdx.x / p.tiles_per_wf;
    const int w0 = (blockIdx.x - n * p.tiles_per_wf) * NW;
    const int L = p.L;
    const int tbase = SGB_SCALE * w0 - 2;         // time of act row 0
    if (p.run_if != nullptr && *p.run_if == 0) return;

    const uint4* const wbase = reinterpret_cast<const uint4*>(p.chunks) + wave * 64 + lane;
    auto wload = [&](int c, int f) -> uint4 { return wbase[((size_t)c * FRAGS_PER_CHUNK + f) * 256]; };
    uint4 wf[2][FRAGS_PER_CHUNK];
#pragma unroll
    for (int f = 0; f < FRAGS_PER_CHUNK; ++f) {
        wf[0][f] = wload(0, f);
        wf[1][f] = wload(1, f);
    }

    for (int i = tid; i < RAWN; i += 256) {
        const int t = tbase - 4 + i;
        raw[i] = (t >= 0 && t < L) ? p.x[(size_t)n * L + t] : 0.f;
    }
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
            store_act4<PREC>(act + row * ROWB, 4 * cq,
                             valid ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    __syncthreads();

    int c = 0;
    if constexpr (SHAPE == 16) {
        // ---- split-fp16 on v_mfma_f32_16x16x32_f16 (stof_common.h "f16x3 SemiGlobalBlock chunks, 16x16x32").  Time stays on
        // the M axis: D[time 4q+e][channel j] of lane (j = lane & 15, q = lane >> 4), the wave's 32 channels are 2 N-tiles,
        // the 80 NW rows are 5 NW M-tiles of 16 -- a pooling window is exactly 5 M-tiles, so the pool is an in-lane max
        // over accumulator registers plus two cross-lane steps (q).  Pipeline unit = the 5 M-tiles of one window against a
        // whole chunk (K = 32): 10 activation fragments from LDS, 30 MFMAs.
        constexpr int MW = SGB_SCALE / 16;                        // M-tiles per pooling window
        const int j16 = lane & 15, q4 = lane >> 4;
        for (int ocb = 0; ocb < 4; ++ocb) {
            floatx4 acc[NW][MW][2];
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[w][m][nt][e] = 0.f;
            constexpr int NUNIT = 10 * NW;                        // 10 chunks (5 taps x 2 halves) x NW windows per output block
            auto aload = [&](uint4 (&a)[MW][2], int uu) {
                const int cc = uu / NW, w = uu % NW;
                const int d = cc >> 1, hh = cc & 1;
                const char* arow = act + (SGB_SCALE * w + j16 + d) * ROWB + 64 * hh + 16 * q4;
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    a[m][0] = ldq(arow + 16 * m * ROWB);
                    a[m][1] = ldq(arow + 16 * m * ROWB + 128);
                }
            };
            auto mfma16 = [](const uint4& a, const uint4& b, floatx4 cacc) {
                return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(a), as_h8(b), cacc, 0, 0, 0);
            };
            auto do_unit = [&](uint4 (&acur)[MW][2], uint4 (&anext)[MW][2], int uu) {
                const int cc = uu / NW, w = uu % NW;
                uint4 (&wq)[FRAGS_PER_CHUNK] = wf[cc & 1];               // fragments: N-tile 0 hi, lo, N-tile 1 hi, lo
                const int c2 = (c + 2 < SGB_NCHUNK) ? c + 2 : 0;          // past the end: harmless reload of chunk 0
                aload(anext, uu + 1 < NUNIT ? uu + 1 : 0);               // the next output block starts with the same rows
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    // hi*hi, lo*hi, hi*lo for both N-tiles, interleaved so that consecutive MFMAs never share an accumulator
                    acc[w][m][0] = mfma16(acur[m][0], wq[0], acc[w][m][0]);
                    acc[w][m][1] = mfma16(acur[m][0], wq[2], acc[w][m][1]);
                    acc[w][m][0] = mfma16(acur[m][1], wq[0], acc[w][m][0]);
                    acc[w][m][1] = mfma16(acur[m][1], wq[2], acc[w][m][1]);
                    acc[w][m][0] = mfma16(acur[m][0], wq[1], acc[w][m][0]);
                    acc[w][m][1] = mfma16(acur[m][0], wq[3], acc[w][m][1]);
                }
                if (w == NW - 1) {
#pragma unroll
                    for (int f = 0; f < FRAGS_PER_CHUNK; ++f) wq[f] = wload(c2, f);
                    ++c;
                }
#pragma unroll
                for (int i = 0; i < MW * 2; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (w == NW - 1) __builtin_amdgcn_sched_group_barrier(0x020, FRAGS_PER_CHUNK, 0);
                __builtin_amdgcn_sched_barrier(0);      // one scheduling region per unit
            };
            uint4 af0[MW][2], af1[MW][2];
            if (ocb == 0) aload(af0, 0);
            static_assert(NUNIT % 2 == 0, "units alternate between two fragment sets");
#pragma unroll
            for (int uu = 0; uu < NUNIT; uu += 2) {
                do_unit(af0, af1, uu);
                do_unit(af1, af0, uu + 1);
            }
            // pool: window w = M-tiles of acc[w]; lane (j16, q4) holds time rows 4 q4 + e of every M-tile for channel j16 of each N-tile
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int oc = 128 * ocb + 32 * wave + 16 * nt + j16;
                const float bias = p.cbias[oc];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    float mval = -INFINITY;
                    if constexpr (ARG) {
                        // training: the window's FIRST maximum and its row (torch's max_pool1d backward routes the gradient there);
                        // bias and leaky ReLU are strictly increasing, so the arg-max of the accumulators is the arg-max of the
                        // activation.  A lane's rows 16 m + 4 q4 + e ascend with (m, e); across the four q4 lanes the larger value
                        // wins and equal values go to the smaller row.
                        int bi = 0;
#pragma unroll
                        for (int m = 0; m < MW; ++m)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = acc[w][m][nt][e];
                                const bool up = v > mval;
                                mval = up ? v : mval;
                                bi = up ? 16 * m + 4 * q4 + e : bi;
                            }
#pragma unroll
                        for (int off = 16; off <= 32; off <<= 1) {
                            const float ov = __shfl_xor(mval, off);
                            const int oi = __shfl_xor(bi, off);
                            const bool take = (ov > mval) || (ov == mval && oi < bi);
                            mval = take ? ov : mval;
                            bi = take ? oi : bi;
                        }
                        if (q4 == 0 && w0 + w < p.P) p.arg[((size_t)n * p.P + w0 + w) * NF_SGB + oc] = (unsigned char)bi;
                    } else {
#pragma unroll
                        for (int m = 0; m < MW; ++m)
#pragma unroll
                            for (int e = 0; e < 4; ++e) mval = fmaxf(mval, acc[w][m][nt][e]);
                        mval = fmaxf(mval, __shfl_xor(mval, 16));
                        mval = fmaxf(mval, __shfl_xor(mval, 32));
                    }
                    mval += bias;
                    mval = mval > 0.f ? mval : 0.01f * mval;
                    if (q4 == 0 && w0 + w < p.P)
                        p.pooled[((size_t)n * p.P + w0 + w) * NF_SGB + oc] = mval;
                }
            }
        }
        return;
    }
    for (int ocb = 0; ocb < 4; ++ocb) {
        floatx16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

        // Pipeline unit = one k-step of a chunk (f16x3: 16 channels, fragments {hi, lo}; fp32: one
        // 8-channel fragment): while the unit's MT x (3 | 4) MFMAs run, the activation fragments
        // (MFMA A operand, time on M) of the next unit are read from LDS into the other register set.
        constexpr int UPC = (PREC == STOF_PREC_FP32) ? 4 : 2;        // units per chunk
        constexpr int FPU = FRAGS_PER_CHUNK / UPC;                    // fragments per unit
        constexpr int NUNIT = 10 * UPC;                               // per 128-channel output block
        auto aload = [&](uint4 (&a)[MT][FPU], int uu) {
            const int cc = uu / UPC, sub = uu % UPC;
            const int d = cc >> 1, hh = cc & 1;
            const char* arow = act + (ln + d) * ROWB;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < FPU; ++i)
                    a[m][i] = ldq(arow + 32 * m * ROWB + act_frag_off<PREC>(sub * FPU + i, hh, lh));
        };
        auto do_unit = [&](uint4 (&acur)[MT][FPU], uint4 (&anext)[MT][FPU], int uu) {
            const int cc = uu / UPC, sub = uu % UPC;
            uint4 (&w)[FRAGS_PER_CHUNK] = wf[cc & 1];
            const int c2 = (c + 2 < SGB_NCHUNK) ? c + 2 : 0;     // past the end: harmless reload of chunk 0
            aload(anext, uu + 1 < NUNIT ? uu + 1 : 0);          // the next output block starts with the same rows
            if constexpr (PREC == STOF_PREC_FP32) {
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mma_fp32(acur[m][0], w[sub], acc[m]);
                w[sub] = wload(c2, sub);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = mma_f16x3(acur[m][0], acur[m][1], w[2 * sub], w[2 * sub + 1], acc[m]);
                w[2 * sub] = wload(c2, 2 * sub);
                w[2 * sub + 1] = wload(c2, 2 * sub + 1);
#pragma unroll
                for (int i = 0; i < MT * FPU; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            if (sub == UPC - 1) ++c;
            __builtin_amdgcn_sched_barrier(0);      // one scheduling region per unit (keeps the igroup solver fast)
        };
        uint4 af0[MT][FPU], af1[MT][FPU];
        if (ocb == 0) aload(af0, 0);
#pragma unroll
        for (int uu = 0; uu < NUNIT; uu += 2) {
            do_unit(af0, af1, uu);
            do_unit(af1, af0, uu + 1);
        }
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
}

// get_maxima_positions in arg-max mode from the tile partials of one sub-batch: one wavefront per waveform.
// Same decision as pick_argmax_kernel (shuffle_picker.hip): with m the row maximum the detections are the positions
// equal to m if m > 0, nothing if m == 0, and for m < 0 only a constant row (or a one-sample window) has any (Q5).
__global__ __launch_bounds__(256) void onsets_finalize_kernel(const OnsetPartial* __restrict__ ws, int nslots, int nb, int r,
                                                              int half, int* __restrict__ counts, int* __restrict__ idx,
                                                              long long idx_cap) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nb) return;
    const OnsetPartial* e = ws + (size_t)row * nslots;
    float m = -INFINITY, lo = INFINITY;
    for (int s = lane; s < nslots; s += 64)
        if (e[s].written) { m = fmaxf(m, e[s].m); lo = fminf(lo, e[s].lo); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); lo = fminf(lo, __shfl_xor(lo, o)); }
    const bool emit = (m > 0.f) || (m < 0.f && (lo == m || half == 0));
    int nout = 0;
    if (emit) {
        const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        int* out = idx + (long long)row * idx_cap;
        for (int s0 = 0; s0 < nslots; s0 += 64) {                // slots are in time order; almost none holds the maximum
            const int sl = s0 + lane;
            unsigned long long match = __ballot(sl < nslots && e[sl].written && e[sl].m == m);
            while (match) {
            const int s = s0 + __builtin_ctzll(match);
            match &= match - 1;
            const OnsetPartial p = e[s];                          // wave-uniform
            for (int q0 = 0; q0 < 16 * r; q0 += 64) {
                const int q = q0 + lane, j = q / r, c = q - j * r;
                const bool hit = q < 16 * r && ((p.eq[c & 3] >> (j + 16 * (c >> 2))) & 1ull);
                const unsigned long long hm = __ballot(hit);
                if (hit) {
                    const long long pos = nout + __builtin_popcountll(hm & lt_mask);
                    if (pos < idx_cap) out[pos] = (p.tw_base + j) * r + c;
                }
                nout += __builtin_popcountll(hm);
            }
            }
        }
    }
    if (lane == 0) counts[row] = nout;
}

constexpr int BODY_S = 192, BODY_RING = 256, BODY_RAWRING = 256;
// The two-pass kernel takes (virtual) waveforms of at least 2 S stream rows: its row arithmetic wraps once per offset without loops
// (the general loops cost the layer set-up ~100 scalar instructions and a division); shorter ones run on the r3 kernel.
inline bool body_p2_rows_ok(int seg_len, int halo) { return seg_len + 2 * halo + GAP >= 2 * BODY_S; }
constexpr int ONSET_MAX_SEGS = 32;       // seg_policy cuts a waveform into at most 2^5 segments

// slots per waveform of the fused picker's partials for rows of L samples cut into nseg segments
constexpr int64_t onset_seg_slots(int64_t seg_len) { return seg_len / 16 + 2; }
constexpr int64_t onset_max_slots(int64_t L) { return L / 16 + 3 * ONSET_MAX_SEGS; }

struct OnsetArgs {                        // non-null ws: stof_forward_onsets
    OnsetPartial* ws;                     // [<= SUB_BATCH][onset_max_slots(L)]
    int32_t* counts;
    int32_t* idx;
    int64_t idx_cap;
    int half;
};
constexpr int64_t SUB_BATCH = 4096;      // rows whose SGB maps share one workspace

constexpr size_t sgb_lds_bytes(int rowf = ROWF) {
    return (size_t)((SGB_SCALE * SGB_NW + 4) * rowf + SGB_SCALE * SGB_NW + 12) * sizeof(float);
}

template <int PREC>
int launch_forward(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y, int64_t N,
                   int64_t L, void* workspace, hipStream_t stream, void* const* events, int32_t* status,
                   const int32_t* run_if, const OnsetArgs* onsets = nullptr) {
    const int force_nseg_log2 = desc->seg_policy > 0 ? desc->seg_policy - 1 : -1;
    const int r = desc->upsample_factor;
    const bool has_sgb = desc->semi_global_scale != 1;
    const int64_t P = L / SGB_SCALE;
    const int64_t rem = L - P * SGB_SCALE;

    // layout of the packed blob (mirrors pack_weights.cpp)
    const float* base = static_cast<const float*>(packed_dev);
    uint64_t off = sizeof(PackedHeader) / sizeof(float);
    const float* c1 = base + off;      off += 64 * 10;
    const float* bias = base + off;    off += 13 * 64;
    const float* body = base + off;    off += (uint64_t)BODY_NCHUNK * BODY_CHUNK_F;
    const float* last16 = nullptr;
    if (PREC == STOF_PREC_F16X3 && r <= 16) { last16 = base + off; off += LAST16_F; }
    static const bool no_last16 = getenv("STOF_NO_LAST16") != nullptr;     // diagnostic A/B switch
    const float* last16_use = no_last16 ? nullptr : last16;
    const float* cbias = base + off;   off += NF_SGB;
    const float* cchunks = base + off; off += (uint64_t)SGB_NCHUNK * SGB_CHUNK_F;
    const float* ew = base + off;      off += 5ull * NF_SGB * NF;
    const float* ebias = base + off;

    // split-fp16 body: 16x16x32 MFMA form unless STOF_BODY16=0 (the packed blob carries the matching fragment order)
    static const bool body16 = PREC == STOF_PREC_F16X3 && body16_enabled();
    // r4: two-pass tile-major layers (body_p2.h) unless STOF_BODY_P2=0 (the r3 chunk-major kernel, kept for A/B runs)
    static const bool body_p2 = body16 && body_p2_enabled();
    constexpr int SHAPE_FAST = PREC == STOF_PREC_F16X3 ? 16 : 32;
    using Lds = BodyLds<BODY_S, BODY_RING, BODY_RAWRING>;
    using Lds16 = BodyLds<BODY_S, BODY_RING, BODY_RAWRING, ROWF16>;
    const size_t body_p2_bytes = Lds16::BYTES + BODY_P2_C1F * sizeof(float);
    const size_t body_lds_bytes = body_p2 ? body_p2_bytes : body16 ? Lds16::BYTES : Lds::BYTES;
    static LdsLimitOnce body_lds, body16_lds, body_p2_lds, sgb_lds;     // one per template instantiation (PREC)
    if (body_p2) {                                   // (+ the r3 kernel: waveforms shorter than 2 S rows go to it, see body_p2_rows_ok)
        if (int st = body_p2_lds.ensure(reinterpret_cast<const void*>(&body_sweep_p2_kernel<BODY_S, BODY_RING, BODY_RAWRING>),
                                        (int)body_p2_bytes)) return st;
        if (int st = body16_lds.ensure(reinterpret_cast<const void*>(&body_sweep_kernel<PREC, BODY_S, BODY_RING, BODY_RAWRING, SHAPE_FAST>),
                                       (int)Lds16::BYTES)) return st;
    } else if (body16) {
        if (int st = body16_lds.ensure(reinterpret_cast<const void*>(&body_sweep_kernel<PREC, BODY_S, BODY_RING, BODY_RAWRING, SHAPE_FAST>),
                                       (int)Lds16::BYTES)) return st;
    } else if (int st = body_lds.ensure(reinterpret_cast<const void*>(&body_sweep_kernel<PREC, BODY_S, BODY_RING, BODY_RAWRING>),
                                        (int)Lds::BYTES)) return st;
    // the SemiGlobalBlock contract kernel follows the body's MFMA shape (and the blob its fragment order)
    static LdsLimitOnce sgb16_lds;
    const size_t sgb_bytes = body16 ? sgb_lds_bytes(ROWF16) : sgb_lds_bytes();
    if (body16) {
        if (int st = sgb16_lds.ensure(reinterpret_cast<const void*>(&sgb_contract_pool_kernel<PREC, SGB_NW, SHAPE_FAST>), (int)sgb_bytes))
            return st;
    } else if (int st = sgb_lds.ensure(reinterpret_cast<const void*>(&sgb_contract_pool_kernel<PREC, SGB_NW>), (int)sgb_bytes))
        return st;
    const int ncu = device_cu_count();

    for (int64_t b0 = 0; b0 < N; b0 += SUB_BATCH) {
        const int64_t nb = (N - b0) < SUB_BATCH ? (N - b0) : SUB_BATCH;
        const float* xb = x + b0 * L;
        float* yb = y ? y + b0 * L * r : nullptr;
        float* pooled = nullptr;
        float* sgb = nullptr;
        const bool ev = events && b0 == 0;
        if (ev) (void)hipEventRecord(static_cast<hipEvent_t>(events[0]), stream);
        if (has_sgb && P > 0) {
            pooled = static_cast<float*>(workspace);
            sgb = pooled + nb * P * NF_SGB;
            SgbParams sp;
            sp.x = xb; sp.pooled = pooled; sp.c1 = c1; sp.cbias = cbias; sp.chunks = cchunks;
            sp.N = (int)nb; sp.L = (int)L; sp.P = (int)P;
            sp.tiles_per_wf = (int)((P + SGB_NW - 1) / SGB_NW);
            sp.run_if = run_if;
            sp.arg = nullptr;
            if (body16)
                hipLaunchKernelGGL((sgb_contract_pool_kernel<PREC, SGB_NW, SHAPE_FAST>), dim3((unsigned)(nb * sp.tiles_per_wf)),
                                   dim3(256), sgb_bytes, stream, sp);
            else
                hipLaunchKernelGGL((sgb_contract_pool_kernel<PREC, SGB_NW>), dim3((unsigned)(nb * sp.tiles_per_wf)),
                                   dim3(256), sgb_bytes, stream, sp);
            if (ev) (void)hipEventRecord(static_cast<hipEvent_t>(events[1]), stream);
            {
                const int st = stof::launch_conv_cl(pooled, ew, ebias, nullptr, nullptr, sgb, 1, nb * (P + 2), NF_SGB, NF, 5,
                                                    /*act = leaky ReLU*/ 2, PREC, (int)(P + 2), (int)P, stream, run_if);
                if (st != STOF_OK) return st;
            }
            if (ev) (void)hipEventRecord(static_cast<hipEvent_t>(events[2]), stream);
        } else if (ev) {
            (void)hipEventRecord(static_cast<hipEvent_t>(events[1]), stream);
            (void)hipEventRecord(static_cast<hipEvent_t>(events[2]), stream);
        }
        BodyParams bp;
        bp.x = xb; bp.sgb = (has_sgb && P > 0) ? sgb : nullptr; bp.y = yb;
        bp.c1 = c1; bp.bias = bias; bp.chunks = body; bp.last16 = last16_use;
        bp.N = (int)nb; bp.L = (int)L; bp.r = r; bp.P = (int)P; bp.rem_half = (int)(rem / 2);
        bp.stamps = nullptr;
        bp.dump = nullptr; bp.dump_stride = 0;
        bp.gin = nullptr; bp.fwd_dump = nullptr;
        bp.status = status;
        bp.run_if = run_if;
        // Cut every waveform into 2^k segments (each swept with +-38 rows of real context, the receptive field of
        // conv1 + 11 x k7 + conv_last) when that shortens the sweep: small batches, batches that do not fill the CUs evenly.
        // k is chosen to minimise the sweep steps of the busiest work-group (it also evens out batches that are not a
        // multiple of the CU count); ties go to fewer segments.
        bp.nseg_log2 = 0;
        {
            int64_t best = -1;
            for (int k = 0; k <= 5; ++k) {
                const int64_t ns = (int64_t)1 << k;
                if (k > 0 && L / ns < BODY_S / 2) break;
                const int64_t lv = (L + ns - 1) / ns + (k > 0 ? 76 : 0);
                const int64_t nvk = nb * ns;
                const int64_t w = nvk < ncu ? nvk : ncu;
                const int64_t per = (nvk + w - 1) / w;
                const int64_t steps = (per * (lv + GAP) - GAP + LAG_LAST + BODY_S - 1) / BODY_S;
                if (best < 0 || steps < best) { best = steps; bp.nseg_log2 = k; }
            }
        }
        int64_t nseg = (int64_t)1 << bp.nseg_log2;
        if (force_nseg_log2 >= 0) { bp.nseg_log2 = force_nseg_log2; nseg = (int64_t)1 << force_nseg_log2; }
        bp.seg_len = (int)((L + nseg - 1) / nseg);
        bp.halo = nseg > 1 ? 38 : 0;
        const int64_t nv = nb * nseg;                     // virtual waveforms
        bp.N = (int)nv;
        bp.onset_ws = nullptr; bp.onset_slots = 0; bp.onset_seg_slots = 0;
        if (onsets != nullptr) {
            bp.onset_seg_slots = (int)onset_seg_slots(bp.seg_len);
            bp.onset_slots = (int)(nseg * bp.onset_seg_slots);
            bp.onset_ws = onsets->ws;
            if (hipMemsetAsync(onsets->ws, 0, (size_t)nb * bp.onset_slots * sizeof(OnsetPartial), stream) != hipSuccess)
                return STOF_ERR_HIP;
        }
#ifdef STOF_STAMPS
        if (workspace) bp.stamps = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + (size_t)(nb * P * (NF_SGB + NF)) * sizeof(float) + 256);
#endif
        // one persistent work-group per CU, each sweeping a contiguous run of waveforms
        int64_t wgs = nv < ncu ? nv : ncu;
        bp.wf_per_wg = (int)((nv + wgs - 1) / wgs);
        wgs = (nv + bp.wf_per_wg - 1) / bp.wf_per_wg;
        if (body_p2 && body_p2_rows_ok(bp.seg_len, bp.halo))
            hipLaunchKernelGGL((body_sweep_p2_kernel<BODY_S, BODY_RING, BODY_RAWRING>), dim3((unsigned)wgs), dim3(256),
                               body_p2_bytes, stream, bp);
        else if (body16)
            hipLaunchKernelGGL((body_sweep_kernel<PREC, BODY_S, BODY_RING, BODY_RAWRING, SHAPE_FAST>), dim3((unsigned)wgs), dim3(256),
                               Lds16::BYTES, stream, bp);
        else
            hipLaunchKernelGGL((body_sweep_kernel<PREC, BODY_S, BODY_RING, BODY_RAWRING>), dim3((unsigned)wgs), dim3(256),
                               body_lds_bytes, stream, bp);
        if (ev) (void)hipEventRecord(static_cast<hipEvent_t>(events[3]), stream);
        if (onsets != nullptr)
            hipLaunchKernelGGL(onsets_finalize_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, stream, onsets->ws,
                               bp.onset_slots, (int)nb, r, onsets->half, onsets->counts + b0, onsets->idx + b0 * onsets->idx_cap,
                               (long long)onsets->idx_cap);
    }
    if (hipGetLastError() != hipSuccess) return STOF_ERR_HIP;
    return STOF_OK;
}

int forward_impl(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y, int64_t N, int64_t L,
                 void* workspace, size_t workspace_bytes, void* stream_, void* const* events, int32_t* status,
                 const int32_t* run_if = nullptr, const OnsetArgs* onsets = nullptr) {
    if (!desc || N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    if (desc->precision != STOF_PREC_FP32 && desc->precision != STOF_PREC_F16X3) return STOF_ERR_UNSUPPORTED;
    if (desc->seg_policy < 0 || desc->seg_policy > 6) return STOF_ERR_UNSUPPORTED;
    const int r = desc->upsample_factor;
    if (r < 1 || r > 64) return STOF_ERR_UNSUPPORTED;
    const bool has_sgb = desc->semi_global_scale != 1;
    if (has_sgb && desc->semi_global_scale != SGB_SCALE) return STOF_ERR_UNSUPPORTED;
    const int64_t P = L / SGB_SCALE;
    if (has_sgb && L > 0 && P == 0) return STOF_ERR_POOL_EMPTY;        // the pooling fails before the add can (L = 79)
    if (has_sgb && ((L - P * SGB_SCALE) & 1)) return STOF_ERR_ODD_SGB_REMAINDER;
    if (N == 0 || L == 0) return STOF_OK;                 // empty batch: nothing to do
    if (!packed_dev || !x || (!y && !onsets)) return STOF_ERR_BAD_ARG;
    if ((L + GAP) * SUB_BATCH > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;   // local stream rows are int32
    if (has_sgb && (!workspace || workspace_bytes < stof_forward_workspace_bytes(desc, N, L)))
        return STOF_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (desc->precision == STOF_PREC_FP32)
        return launch_forward<STOF_PREC_FP32>(desc, packed_dev, x, y, N, L, workspace, stream, events, status, run_if, onsets);
    return launch_forward<STOF_PREC_F16X3>(desc, packed_dev, x, y, N, L, workspace, stream, events, status, run_if, onsets);
}

}  // namespace

extern "C" size_t stof_forward_workspace_bytes(const stof_net_desc* desc, int64_t N, int64_t L) {
    if (!desc || N <= 0 || L <= 0 || desc->semi_global_scale == 1) return 0;
    const int64_t nb = N < SUB_BATCH ? N : SUB_BATCH;
    const int64_t P = L / SGB_SCALE;
#ifdef STOF_STAMPS
    return (size_t)(nb * P * (NF_SGB + NF)) * sizeof(float) + 256 + 256 * 4 * 8 * 8 + 64;
#else
    return (size_t)(nb * P * (NF_SGB + NF)) * sizeof(float) + 256;
#endif
}

extern "C" int stof_forward(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y,
                            int64_t N, int64_t L, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_impl(desc, packed_dev, x, y, N, L, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int stof_forward_checked(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y,
                                    int64_t N, int64_t L, void* workspace, size_t workspace_bytes, void* stream,
                                    int32_t* status_dev) {
    if (!status_dev) return STOF_ERR_BAD_ARG;
    return forward_impl(desc, packed_dev, x, y, N, L, workspace, workspace_bytes, stream, nullptr, status_dev);
}

extern "C" int stof_forward_events(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y,
                                   int64_t N, int64_t L, void* workspace, size_t workspace_bytes, void* stream,
                                   void* const* events, int32_t* status_dev) {
    if (!events) return STOF_ERR_BAD_ARG;
    for (int e = 0; e < STOF_FORWARD_EVENTS; ++e)
        if (!events[e]) return STOF_ERR_BAD_ARG;
    return forward_impl(desc, packed_dev, x, y, N, L, workspace, workspace_bytes, stream, events, status_dev);
}

// The picker fused into the sweep: workspace = the forward's workspace followed by the tile partials of one sub-batch.
static size_t onsets_partials_offset(const stof_net_desc* desc, int64_t N, int64_t L) {
    return (stof_forward_workspace_bytes(desc, N, L) + 255) / 256 * 256;
}

extern "C" size_t stof_forward_onsets_workspace_bytes(const stof_net_desc* desc, int64_t N, int64_t L) {
    if (!desc || N <= 0 || L <= 0) return 0;
    const int64_t nb = N < SUB_BATCH ? N : SUB_BATCH;
    return onsets_partials_offset(desc, N, L) + (size_t)nb * onset_max_slots(L) * sizeof(OnsetPartial);
}

extern "C" int stof_forward_onsets(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y, int64_t N,
                                   int64_t L, int32_t window_size, int32_t* counts, int32_t* idx, int64_t idx_cap,
                                   void* workspace, size_t workspace_bytes, void* stream, int32_t* status_dev) {
    if (!desc || !counts || (!idx && idx_cap > 0) || idx_cap < 0 || window_size < 0) return STOF_ERR_BAD_ARG;
    // the fused picker lives in the 16-channel conv_last tile of the split-fp16 sweep
    if (desc->precision != STOF_PREC_F16X3 || desc->upsample_factor > 16 || L < 32 || getenv("STOF_NO_LAST16") != nullptr)
        return STOF_ERR_UNSUPPORTED;
    if (N > 0 && L > 0 && (!workspace || workspace_bytes < stof_forward_onsets_workspace_bytes(desc, N, L))) return STOF_ERR_WORKSPACE;
    OnsetArgs oa;
    oa.ws = reinterpret_cast<OnsetPartial*>(static_cast<char*>(workspace) + onsets_partials_offset(desc, N, L));
    oa.counts = counts; oa.idx = idx; oa.idx_cap = idx_cap;
    oa.half = (window_size / 2 * 2 + 1 - 1) / 2;                 // utils/mask2samples.py:7-8
    return forward_impl(desc, packed_dev, x, y, N, L, workspace, workspace_bytes, stream, nullptr, status_dev, nullptr, &oa);
}

extern "C" int stof_forward_auto(const stof_net_desc* desc, const void* packed_f16x3_dev, const void* packed_fp32_dev,
                                 const float* x, float* y, int64_t N, int64_t L, void* workspace, size_t workspace_bytes,
                                 void* stream, int32_t* status_dev, void* const* events) {
    if (!desc || !status_dev) return STOF_ERR_BAD_ARG;
    if (events)
        for (int e = 0; e < STOF_FORWARD_EVENTS; ++e)
            if (!events[e]) return STOF_ERR_BAD_ARG;
    stof_net_desc d16 = *desc, d32 = *desc;
    d16.precision = STOF_PREC_F16X3;
    d32.precision = STOF_PREC_FP32;
    if (N > 0 && L > 0 && hipMemsetAsync(status_dev, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)) != hipSuccess)
        return STOF_ERR_HIP;
    const int st = forward_impl(&d16, packed_f16x3_dev, x, y, N, L, workspace, workspace_bytes, stream, events, status_dev);
    if (st != STOF_OK) return st;
    // exact-fp32 re-run of the whole call, gated on the device by the range-guard word: its three launches return at
    // once while *status_dev == 0, so the common case costs three empty launches and no host synchronisation
    return forward_impl(&d32, packed_fp32_dev, x, y, N, L, workspace, workspace_bytes, stream, nullptr, nullptr, status_dev);
}

// ----------------------------------------------------------------------------------
// Training forward on the sweep (main.py:221 in train mode): conv2 .. conv12 + conv_last of the split-fp16
// training step run as ONE body sweep that also writes every layer's output to HBM for the backward pass, instead of twelve
// layer-by-layer launches.  The parameters change every step, so the sweep's operand blob is packed ON THE DEVICE.
// ----------------------------------------------------------------------------------
namespace {

struct SweepPackArgs {
    const float* p[26];       // conv1.w, conv1.b, conv2.w, conv2.b, ..., conv12.b, conv_last.w, conv_last.b (stof_pack_weights order)
    float* blob;              // header-less: [c1 640][bias 832][body BODY_NCHUNK * BODY_CHUNK_F][last16 LAST16_F if r <= 16]
    int r;
};

__global__ __launch_bounds__(256) void sweep_pack_kernel(const SweepPackArgs a) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    const long long n_c1 = 640, n_bias = 13 * 64, n_body = (long long)BODY_NCHUNK * BODY_CHUNK_F * 2;   // body counted in fp16 halves
    const long long n_last = a.r <= 16 ? (long long)LAST16_F * 2 : 0;
    if (i < n_c1) {
        const int c = (int)(i / 10), t = (int)(i % 10);
        a.blob[i] = t < 9 ? a.p[0][c * 9 + t] : a.p[1][c];
        return;
    }
    long long k = i - n_c1;
    if (k < n_bias) {
        const int j = (int)(k / 64), c = (int)(k % 64);
        float v = 0.f;
        if (j >= 1 && j <= 11) v = a.p[3 + 2 * (j - 1)][c];
        else if (j == 12 && c < a.r) v = a.p[25][c];
        a.blob[n_c1 + k] = v;
        return;
    }
    k -= n_bias;
    _Float16* const body = reinterpret_cast<_Float16*>(a.blob + n_c1 + n_bias);
    if (k < n_body) {
        // half index -> (chunk, frag, block, lane, e), the order of pack_chunk16 (pack_weights.cpp)
        const int e = (int)(k & 7), lane = (int)((k >> 3) & 63), blk = (int)((k >> 9) & 1), frag = (int)((k >> 10) & 3);
        const int c = (int)(k >> 12);
        const int j = c < 11 * BODY_CHUNKS_K7 ? 1 + c / BODY_CHUNKS_K7 : 12;
        const int cc = c < 11 * BODY_CHUNKS_K7 ? c % BODY_CHUNKS_K7 : c - 11 * BODY_CHUNKS_K7;
        const int K = j == 12 ? 3 : 7, co = j == 12 ? a.r : NF;
        const int tap = cc >> 1, hh = cc & 1, m = frag >> 1, part = frag & 1, i16 = lane & 15, q = lane >> 4;
        const int o = body16_out_channel(blk, m, i16), ch = 32 * hh + 8 * q + e;
        const float* w = j == 12 ? a.p[24] : a.p[2 + 2 * (j - 1)];
        const float v = o < co ? w[((size_t)o * NF + ch) * K + tap] : 0.f;
        const _Float16 hi = (_Float16)v;
        body[k] = part == 0 ? hi : (_Float16)(v - (float)hi);
        return;
    }
    k -= n_body;
    if (k < n_last) {
        // [6 chunks][hi | lo][64 lanes][8]: conv_last as 16x16x32 A operands, rows >= r zero
        const int e = (int)(k & 7), lane = (int)((k >> 3) & 63), part = (int)((k >> 9) & 1), cc = (int)(k >> 10);
        const int tap = cc >> 1, hh = cc & 1, o = lane & 15, ch = 32 * hh + 8 * (lane >> 4) + e;
        const float v = o < a.r ? a.p[24][((size_t)o * NF + ch) * 3 + tap] : 0.f;
        const _Float16 hi = (_Float16)v;
        reinterpret_cast<_Float16*>(a.blob + n_c1 + n_bias + (long long)BODY_NCHUNK * BODY_CHUNK_F)[k] = part == 0 ? hi : (_Float16)(v - (float)hi);
    }
}

inline size_t sweep_blob_floats(int r) {
    return 640 + 13 * 64 + (size_t)BODY_NCHUNK * BODY_CHUNK_F + (r <= 16 ? LAST16_F : 0);
}

}  // namespace

extern "C" size_t stof_train_sweep_blob_bytes(const stof_net_desc* desc) {
    if (!desc || desc->upsample_factor < 1 || desc->upsample_factor > 64) return 0;
    return sweep_blob_floats(desc->upsample_factor) * sizeof(float);
}

extern "C" int stof_train_sweep_pack(const stof_net_desc* desc, const float* const* params_dev, void* blob_dev, void* stream) {
    if (!desc || !params_dev || !blob_dev) return STOF_ERR_BAD_ARG;
    if (desc->upsample_factor < 1 || desc->upsample_factor > 64) return STOF_ERR_UNSUPPORTED;
    SweepPackArgs a;
    for (int i = 0; i < 26; ++i) {
        if (!params_dev[i]) return STOF_ERR_BAD_ARG;
        a.p[i] = params_dev[i];
    }
    a.blob = static_cast<float*>(blob_dev);
    a.r = desc->upsample_factor;
    const long long total = 640 + 13 * 64 + (long long)BODY_NCHUNK * BODY_CHUNK_F * 2 + (a.r <= 16 ? (long long)LAST16_F * 2 : 0);
    hipLaunchKernelGGL(sweep_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" size_t stof_train_sweep_dump_floats(int64_t N, int64_t L) {
    if (N <= 0 || L <= 0) return 0;
    return (size_t)12 * N * L * NF + 1024;
}

namespace {
// shared launch geometry of the training sweeps (as the inference launch: segments keep the CUs busy for small batches)
int train_sweep_geometry(const stof_net_desc* desc, BodyParams& bp, int64_t N, int64_t L, int64_t* wgs_out) {
    const int ncu = device_cu_count();
    bp.nseg_log2 = 0;
    int64_t best = -1;
    for (int k = 0; k <= 5; ++k) {
        const int64_t ns = (int64_t)1 << k;
        if (k > 0 && L / ns < BODY_S / 2) break;
        const int64_t lv = (L + ns - 1) / ns + (k > 0 ? 76 : 0);
        const int64_t nvk = N * ns;
        const int64_t w = nvk < ncu ? nvk : ncu;
        const int64_t per = (nvk + w - 1) / w;
        const int64_t steps = (per * (lv + GAP) - GAP + LAG_LAST + BODY_S - 1) / BODY_S;
        if (best < 0 || steps < best) { best = steps; bp.nseg_log2 = k; }
    }
    if (desc->seg_policy > 0) bp.nseg_log2 = desc->seg_policy - 1;
    const int64_t nseg = (int64_t)1 << bp.nseg_log2;
    bp.seg_len = (int)((L + nseg - 1) / nseg);
    bp.halo = nseg > 1 ? 38 : 0;
    const int64_t nv = N * nseg;
    if ((int64_t)(bp.seg_len + 2 * bp.halo + GAP) * nv > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    bp.N = (int)nv;
    int64_t wgs = nv < ncu ? nv : ncu;
    bp.wf_per_wg = (int)((nv + wgs - 1) / wgs);
    *wgs_out = (nv + bp.wf_per_wg - 1) / bp.wf_per_wg;
    return STOF_OK;
}
}  // namespace

#ifdef STOF_STAMPS
// diagnostic builds only: cycle stamps of the training sweeps (slot 0 = forward, 1 = backward), read back by stof_debug_train_stamps
static unsigned long long* train_stamp_last[2] = {nullptr, nullptr};
static unsigned long long* train_stamp_buf(int which) {
    static unsigned long long* buf[2] = {nullptr, nullptr};
    if (!buf[which] && hipMalloc(&buf[which], 256 * 4 * 8 * 8) != hipSuccess) return nullptr;
    (void)hipMemset(buf[which], 0, 256 * 4 * 8 * 8);
    return train_stamp_last[which] = buf[which];
}
extern "C" int stof_debug_train_stamps(int which, unsigned long long* host_out) {
    if (which < 0 || which > 1 || !train_stamp_last[which]) return STOF_ERR_BAD_ARG;
    return hipMemcpy(host_out, train_stamp_last[which], 256 * 4 * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}
#endif
static int train_sweep_impl(const stof_net_desc* desc, const void* blob_dev, const float* x, const float* sgb_expand,
                            float* dump, float* y, int64_t N, int64_t L, void* stream_, bool split) {
    if (!desc || N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    const int r = desc->upsample_factor;
    if (r < 1 || r > 64) return STOF_ERR_UNSUPPORTED;
    const bool has_sgb = desc->semi_global_scale != 1;
    if (has_sgb && desc->semi_global_scale != SGB_SCALE) return STOF_ERR_UNSUPPORTED;
    const int64_t P = has_sgb ? L / SGB_SCALE : 0;
    if (has_sgb && L > 0 && P == 0) return STOF_ERR_POOL_EMPTY;
    if (has_sgb && ((L - P * SGB_SCALE) & 1)) return STOF_ERR_ODD_SGB_REMAINDER;
    if (N == 0 || L == 0) return STOF_OK;
    if (!blob_dev || !x || !dump || !y || (has_sgb && !sgb_expand)) return STOF_ERR_BAD_ARG;
    if ((L + GAP) * N > 0x7fffffffLL || N * L * NF > 0x7fffffffffLL) return STOF_ERR_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using Lds16 = BodyLds<BODY_S, BODY_RING, BODY_RAWRING, ROWF16>;
    static const bool p2 = body_p2_enabled();
    if (split && !p2) return STOF_ERR_UNSUPPORTED;              // split-row dumps exist in the two-pass kernel only
    auto kernel = split ? &body_sweep_p2_kernel<BODY_S, BODY_RING, BODY_RAWRING, true, false, true>
                  : p2  ? &body_sweep_p2_kernel<BODY_S, BODY_RING, BODY_RAWRING, true, false>
                        : &body_sweep_kernel<STOF_PREC_F16X3, BODY_S, BODY_RING, BODY_RAWRING, 16, true, false>;
    const size_t lds_bytes = Lds16::BYTES + (p2 ? BODY_P2_C1F * sizeof(float) : 0);
    static LdsLimitOnce lds[2];
    if (int st = lds[split ? 1 : 0].ensure(reinterpret_cast<const void*>(kernel), (int)lds_bytes)) return st;
    const float* base = static_cast<const float*>(blob_dev);
    BodyParams bp;
    bp.x = x; bp.sgb = has_sgb ? sgb_expand : nullptr; bp.y = y;
    bp.c1 = base; bp.bias = base + 640; bp.chunks = base + 640 + 13 * 64;
    bp.last16 = r <= 16 ? bp.chunks + (size_t)BODY_NCHUNK * BODY_CHUNK_F : nullptr;
    bp.L = (int)L; bp.r = r; bp.P = (int)P; bp.rem_half = (int)((L - P * SGB_SCALE) / 2);
    bp.stamps = nullptr; bp.status = nullptr; bp.run_if = nullptr;
    bp.onset_ws = nullptr; bp.onset_slots = 0; bp.onset_seg_slots = 0;
    bp.dump = dump; bp.dump_stride = (long long)N * L * NF;
    bp.gin = nullptr; bp.fwd_dump = nullptr;
#ifdef STOF_STAMPS
    bp.stamps = train_stamp_buf(0);
#endif
    int64_t wgs = 0;
    if (int st = train_sweep_geometry(desc, bp, N, L, &wgs)) return st;
    if (p2 && !body_p2_rows_ok(bp.seg_len, bp.halo)) {            // short waveforms: the r3 kernel (fp32 dumps only)
        if (split) return STOF_ERR_UNSUPPORTED;
        auto k3 = &body_sweep_kernel<STOF_PREC_F16X3, BODY_S, BODY_RING, BODY_RAWRING, 16, true, false>;
        static LdsLimitOnce lds3;
        if (int st = lds3.ensure(reinterpret_cast<const void*>(k3), (int)Lds16::BYTES)) return st;
        hipLaunchKernelGGL(k3, dim3((unsigned)wgs), dim3(256), Lds16::BYTES, stream, bp);
        return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)wgs), dim3(256), lds_bytes, stream, bp);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_sweep(const stof_net_desc* desc, const void* blob_dev, const float* x, const float* sgb_expand,
                                float* dump, float* y, int64_t N, int64_t L, void* stream) {
    return train_sweep_impl(desc, blob_dev, x, sgb_expand, dump, y, N, L, stream, false);
}
extern "C" int stof_train_sweep_split(const stof_net_desc* desc, const void* blob_dev, const float* x, const float* sgb_expand,
                                      float* dump, float* y, int64_t N, int64_t L, void* stream) {
    return train_sweep_impl(desc, blob_dev, x, sgb_expand, dump, y, N, L, stream, true);
}

// ---- training forward of the SemiGlobalBlock's contracting path: relu(conv1) -> contract conv -> lrelu -> max-pool(80) fused as in
// inference (the [N, L, 512] activation never reaches HBM), plus the pool's arg-max for the backward pass
namespace {
struct SgbPackArgs {
    const float* c1w; const float* c1b; const float* cw; const float* cb;     // conv1 (64,1,9), (64); contract_conv (512,64,5), (512)
    float* blob;              // [c1 640][cbias 512][SGB_NCHUNK * SGB_CHUNK_F]
};
__global__ __launch_bounds__(256) void sgb_pack_kernel(const SgbPackArgs a) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    const long long n_c1 = 640, n_cb = NF_SGB, n_body = (long long)SGB_NCHUNK * SGB_CHUNK_F * 2;
    if (i < n_c1) {
        const int c = (int)(i / 10), t = (int)(i % 10);
        a.blob[i] = t < 9 ? a.c1w[c * 9 + t] : a.c1b[c];
        return;
    }
    long long k = i - n_c1;
    if (k < n_cb) { a.blob[n_c1 + k] = a.cb[k]; return; }
    k -= n_cb;
    if (k >= n_body) return;
    // half index -> (chunk = (ocb, tap, hh), frag, tile, lane, e), the order of pack_chunk16_sgb (pack_weights.cpp)
    const int e = (int)(k & 7), lane = (int)((k >> 3) & 63), tile = (int)((k >> 9) & 3), frag = (int)((k >> 11) & 3);
    const int c = (int)(k >> 13);
    const int hh = c & 1, tap = (c >> 1) % 5, ocb = c / 10;
    const int nt = frag >> 1, part = frag & 1, j = lane & 15, q = lane >> 4;
    const int o = 128 * ocb + 32 * tile + 16 * nt + j;
    const float v = a.cw[((size_t)o * NF + 32 * hh + 8 * q + e) * 5 + tap];
    const _Float16 hi = (_Float16)v;
    reinterpret_cast<_Float16*>(a.blob + n_c1 + n_cb)[k] = part == 0 ? hi : (_Float16)(v - (float)hi);
}
}  // namespace

extern "C" size_t stof_train_sgb_blob_bytes(void) { return (640 + NF_SGB + (size_t)SGB_NCHUNK * SGB_CHUNK_F) * sizeof(float); }

extern "C" int stof_train_sgb_contract_pool(const float* conv1_w, const float* conv1_b, const float* contract_w, const float* contract_b,
                                            void* blob_dev, const float* x, float* pooled, uint8_t* arg, int64_t N, int64_t L,
                                            void* stream_) {
    if (N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    const int64_t P = L / SGB_SCALE;
    if (L > 0 && P == 0) return STOF_ERR_POOL_EMPTY;
    if (N == 0 || L == 0) return STOF_OK;
    if (!conv1_w || !conv1_b || !contract_w || !contract_b || !blob_dev || !x || !pooled || !arg) return STOF_ERR_BAD_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SgbPackArgs a;
    a.c1w = conv1_w; a.c1b = conv1_b; a.cw = contract_w; a.cb = contract_b; a.blob = static_cast<float*>(blob_dev);
    const long long total = 640 + NF_SGB + (long long)SGB_NCHUNK * SGB_CHUNK_F * 2;
    hipLaunchKernelGGL(sgb_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    auto kernel = &sgb_contract_pool_kernel<STOF_PREC_F16X3, SGB_NW, 16, true>;
    static LdsLimitOnce lds;
    if (int st = lds.ensure(reinterpret_cast<const void*>(kernel), (int)sgb_lds_bytes(ROWF16))) return st;
    SgbParams sp;
    sp.x = x; sp.pooled = pooled; sp.c1 = a.blob; sp.cbias = a.blob + 640; sp.chunks = a.blob + 640 + NF_SGB;
    sp.N = (int)N; sp.L = (int)L; sp.P = (int)P;
    sp.tiles_per_wf = (int)((P + SGB_NW - 1) / SGB_NW);
    sp.run_if = nullptr; sp.arg = arg;
    if (N * sp.tiles_per_wf > 0x7fffffffLL) return STOF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(N * sp.tiles_per_wf)), dim3(256), sgb_lds_bytes(ROWF16), stream, sp);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

// ---- backward sweep: the eleven data-gradient convolutions conv12^T .. conv2^T of the training step in one launch
namespace {
struct SweepPackBwdArgs {
    const float* w[11];       // conv2.weight .. conv12.weight (forward order), each (64, 64, 7)
    float* blob;              // [c1 640 unused][bias 832 zeros][BODY_NCHUNK * BODY_CHUNK_F]
};
__global__ __launch_bounds__(256) void sweep_pack_bwd_kernel(const SweepPackBwdArgs a) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    const long long n_head = 640 + 13 * 64, n_body = (long long)BODY_NCHUNK * BODY_CHUNK_F * 2;
    if (i < n_head) { a.blob[i] = 0.f; return; }
    const long long k = i - n_head;
    if (k >= n_body) return;
    _Float16* const body = reinterpret_cast<_Float16*>(a.blob + n_head);
    const int e = (int)(k & 7), lane = (int)((k >> 3) & 63), blk = (int)((k >> 9) & 1), frag = (int)((k >> 10) & 3);
    const int c = (int)(k >> 12);
    if (c >= 11 * BODY_CHUNKS_K7) { body[k] = (_Float16)0.f; return; }
    const int j = 1 + c / BODY_CHUNKS_K7, cc = c % BODY_CHUNKS_K7;      // sweep layer j = conv(13 - j)^T
    const int tap = cc >> 1, hh = cc & 1, m = frag >> 1, part = frag & 1, i16 = lane & 15, q = lane >> 4;
    const int arow = body16_out_channel(blk, m, i16);                    // output of the transposed conv = input channel c of conv(13 - j)
    const int b = 32 * hh + 8 * q + e;                                   // contraction index = its output channel o
    const float v = a.w[11 - j][((size_t)b * NF + arow) * 7 + (6 - tap)];
    const _Float16 hi = (_Float16)v;
    body[k] = part == 0 ? hi : (_Float16)(v - (float)hi);
}
}  // namespace

extern "C" int stof_train_sweep_bwd_pack(const float* const* conv_weights_dev, void* blob_dev, void* stream) {
    if (!conv_weights_dev || !blob_dev) return STOF_ERR_BAD_ARG;
    SweepPackBwdArgs a;
    for (int i = 0; i < 11; ++i) {
        if (!conv_weights_dev[i]) return STOF_ERR_BAD_ARG;
        a.w[i] = conv_weights_dev[i];
    }
    a.blob = static_cast<float*>(blob_dev);
    const long long total = 640 + 13 * 64 + (long long)BODY_NCHUNK * BODY_CHUNK_F * 2;
    hipLaunchKernelGGL(sweep_pack_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

static int train_sweep_bwd_impl(const stof_net_desc* desc, const void* blob_dev, const float* g6, const float* fwd_dump,
                                float* dump, int64_t N, int64_t L, void* stream_, bool split) {
    if (!desc || N < 0 || L < 0) return STOF_ERR_BAD_ARG;
    if (N == 0 || L == 0) return STOF_OK;
    if (!blob_dev || !g6 || !fwd_dump || !dump) return STOF_ERR_BAD_ARG;
    if ((L + GAP) * N > 0x7fffffffLL || N * L * NF > 0x7fffffffffLL) return STOF_ERR_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using Lds16 = BodyLds<BODY_S, BODY_RING, BODY_RAWRING, ROWF16>;
    static const bool p2 = body_p2_enabled();
    if (split && !p2) return STOF_ERR_UNSUPPORTED;
    auto kernel = split ? &body_sweep_p2_kernel<BODY_S, BODY_RING, BODY_RAWRING, true, true, true>
                  : p2  ? &body_sweep_p2_kernel<BODY_S, BODY_RING, BODY_RAWRING, true, true>
                        : &body_sweep_kernel<STOF_PREC_F16X3, BODY_S, BODY_RING, BODY_RAWRING, 16, true, true>;
    const size_t lds_bytes = Lds16::BYTES + (p2 ? BODY_P2_C1F * sizeof(float) : 0);
    static LdsLimitOnce lds[2];
    if (int st = lds[split ? 1 : 0].ensure(reinterpret_cast<const void*>(kernel), (int)lds_bytes)) return st;
    const float* base = static_cast<const float*>(blob_dev);
    BodyParams bp;
    bp.x = nullptr; bp.sgb = nullptr; bp.y = nullptr;
    bp.c1 = base; bp.bias = base + 640; bp.chunks = base + 640 + 13 * 64; bp.last16 = nullptr;
    bp.L = (int)L; bp.r = 1; bp.P = 0; bp.rem_half = 0;
    bp.stamps = nullptr; bp.status = nullptr; bp.run_if = nullptr;
    bp.onset_ws = nullptr; bp.onset_slots = 0; bp.onset_seg_slots = 0;
    bp.dump = dump; bp.dump_stride = (long long)N * L * NF;
    bp.gin = g6; bp.fwd_dump = fwd_dump;
#ifdef STOF_STAMPS
    bp.stamps = train_stamp_buf(1);
#endif
    int64_t wgs = 0;
    if (int st = train_sweep_geometry(desc, bp, N, L, &wgs)) return st;
    if (p2 && !body_p2_rows_ok(bp.seg_len, bp.halo)) {            // short waveforms: the r3 kernel (fp32 dumps only)
        if (split) return STOF_ERR_UNSUPPORTED;
        auto k3 = &body_sweep_kernel<STOF_PREC_F16X3, BODY_S, BODY_RING, BODY_RAWRING, 16, true, true>;
        static LdsLimitOnce lds3;
        if (int st = lds3.ensure(reinterpret_cast<const void*>(k3), (int)Lds16::BYTES)) return st;
        hipLaunchKernelGGL(k3, dim3((unsigned)wgs), dim3(256), Lds16::BYTES, stream, bp);
        return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)wgs), dim3(256), lds_bytes, stream, bp);
    return hipGetLastError() == hipSuccess ? STOF_OK : STOF_ERR_HIP;
}

extern "C" int stof_train_sweep_bwd(const stof_net_desc* desc, const void* blob_dev, const float* g6, const float* fwd_dump,
                                    float* dump, int64_t N, int64_t L, void* stream) {
    return train_sweep_bwd_impl(desc, blob_dev, g6, fwd_dump, dump, N, L, stream, false);
}
extern "C" int stof_train_sweep_bwd_split(const stof_net_desc* desc, const void* blob_dev, const float* g6, const float* fwd_dump,
                                          float* dump, int64_t N, int64_t L, void* stream) {
    return train_sweep_bwd_impl(desc, blob_dev, g6, fwd_dump, dump, N, L, stream, true);
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
