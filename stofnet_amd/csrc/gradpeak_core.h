// grad_peak_detect (models/gradpeak.py:8-68) for one row per wavefront -- device code shared by gradpeak.hip's stand-alone
// kernels (envelope rows in HBM) and by the fused toa_detect kernels (envelope rows in LDS, straight out of the inverse
// FFT):
//
//   gradient  torch.gradient(env, spacing = g)  (:14): central differences / (2 g), one-sided / g at both ends; every
//             value is the reference's float (IEEE division, or its correctly rounded constant-divisor form)
//   blur      zero padded correlation with the 2 rad + 1 Gaussian taps (:15, :89-96), fmaf chain in tap order
//   edges     rising edges of (blur > th) and (blur < -th/4) (:23-30) as 64-bit ballot masks; an edge sits at the
//             last-false sample (diff == 1 at i means flag[i] = 0, flag[i + 1] = 1)
//   pairing   every falling-slope edge `am` takes the nearest rising-slope edge `ap` <= am, gate
//             ival_min < am - ap < ival_max, first am per distinct ap (:42-60)
//
// The smoothed gradient never leaves registers: gradients go through a small buffer in LDS (the blur needs 2 rad + 1
// neighbours), the two comparison flags of 64 consecutive samples are two ballots.  Two streamers:
//   stream_blocks  (r3, every kernel): four 64-sample words per iteration, flag words collected 64 at a time and paired
//                  together (pair_batch) or stored for a later pairing (pair_stored_words)
//   stream_words   (r2): one word per iteration; kept for the pre-pass that stores the smoothed gradient
//                  (stof_gradpeak_moments_store)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "fft_small.h"      // wave_lds_sync

namespace stof_gp {

constexpr int MAXRAD = 96;                      // Gaussian radius limit (sigma = (2 g - 1) / 6, rf_scale_factor <= 115)
constexpr int TAPS_LDS = 2 * MAXRAD + 8;        // floats of the zero-padded tap image in LDS (a multiple of 8)

// copy the taps into their LDS image (all threads of the work-group; the caller synchronises)
__device__ __forceinline__ void stage_taps(float* __restrict__ tp, const float* __restrict__ taps, int radius, int tid, int T) {
    const int ntaps = 2 * radius + 1;
    for (int i = tid; i < TAPS_LDS; i += T) tp[i] = i < ntaps ? taps[i] : 0.f;
}

struct Config {
    int L;                   // samples per row
    float spacing;           // g
    int radius;              // rad
    float th_pos, th_neg;
    int ival_min, ival_max;
    long long cap;           // echoes kept per row in `echoes`
    long long echo_max;      // > 0: also write the echo_max reduction (models/gradpeak.py:107-114) to `reduced`
};

// ring entries for a radius: the blur of 64 samples reads 64 + 2 rad gradients.  Every gradient is stored twice, at
// slot and slot + ring_entries, so the 2 rad + 1 reads of a sample are `base + j` with no wrap arithmetic (one
// ds_read with an immediate offset per tap); a row's ring therefore takes 2 * ring_entries floats.
__host__ __device__ constexpr int ring_entries(int radius) { return radius <= 32 ? 128 : 256; }
__host__ __device__ constexpr int ring_floats(int radius) { return 2 * ring_entries(radius); }

struct RowState {
    int last_ap = -1;          // most recent rising-slope edge seen so far (carry across words)
    int done = 0;              // last_ap already has its peak (first am per distinct ap, :58-59)
    int nout = 0;
    int any_ap = 0, any_am = 0;
    unsigned long long P = 0, M = 0, V = 0;      // flags / validity of the previous word
};

// One word of the pairing: samples base .. base + 63, rising-slope edges EP, falling-slope edges EM (ballot words, so
// everything here is wave-uniform and runs on the scalar unit).
// Reference (:42-60): every falling-slope edge `am` takes the nearest rising-slope edge `ap` <= am, the gate keeps
// ival_min < am - ap < ival_max, and per distinct ap the FIRST surviving am is kept.  Read from the ap side: an onset ap
// owns the stretch [ap, next ap) and its peak is the first EM edge of that stretch inside the gate.  Noisy rows have a
// falling-slope edge every dozen samples but only a few onsets, so the work is per onset, not per edge: a word without
// a pending onset and without a new one costs one scalar test.  (The reference's 2**32 sentinel, :43 / Q8, maps a peak
// without a preceding onset to the first onset; such a peak is dropped here as it is there for any ival_min >= 0.)
__device__ __forceinline__ int msb64(unsigned long long v) { return 63 - __builtin_clzll(v); }

// The same pairing for a word crowded with onsets (a threshold near zero makes every noise wiggle an onset): one lane
// per sample, each falling-slope edge looks up its onset with mask arithmetic and a ballot prefix orders the survivors.
// Same state as the per-onset form: `done` <=> the last surviving candidate's onset is last_ap.
template <class EnvAt>
__device__ __forceinline__ void pair_word_dense(RowState& st, int base, unsigned long long EP, unsigned long long EM, int lane,
                                                const Config& cf, float* __restrict__ out, EnvAt env_at) {
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long le_mask = lt_mask | (1ull << lane);
    const int i = base + lane;
    const bool em = (EM >> lane) & 1ull;
    const unsigned long long below = EP & le_mask;             // nearest preceding (<=) onset of this lane's edge (:42-45)
    const int ap = below ? (base + msb64(below)) : st.last_ap;
    const int gap = i - ap;
    const bool valid = em && (ap >= 0) && (gap > cf.ival_min) && (gap < cf.ival_max);     // :48-49
    const unsigned long long vm = __ballot(valid);
    const unsigned long long vbelow = vm & lt_mask;
    const int prev_ap_lane = __shfl(ap, vbelow ? msb64(vbelow) : 0);
    const int prev_ap = vbelow ? prev_ap_lane : (st.done ? st.last_ap : -2);   // onset of the previous surviving candidate
    const bool keep = valid && (ap != prev_ap);                // first am per distinct ap (:58-59)
    const unsigned long long km = __ballot(keep);
    if (keep) {
        const long long pos = st.nout + __builtin_popcountll(km & lt_mask);
        if (pos < cf.cap) {
            out[3 * pos + 0] = (float)ap;
            out[3 * pos + 1] = (float)i;
            out[3 * pos + 2] = env_at(i);                      // data[i, am] (:66)
        }
    }
    st.nout += __builtin_popcountll(km);
    // (readlane, not a shuffle: the result must stay wave-uniform for the compiler, or the whole pairing state turns into
    // vector registers and every test on it into an exec-masked region)
    const int last_valid_ap = vm ? __builtin_amdgcn_readlane(ap, msb64(vm)) : (st.done ? st.last_ap : -2);
    if (EP) st.last_ap = base + msb64(EP);
    st.done = (last_valid_ap == st.last_ap);
}

template <class EnvAt>
__device__ __forceinline__ void pair_word(RowState& st, int base, unsigned long long EP, unsigned long long EM, int lane,
                                          const Config& cf, float* __restrict__ out, EnvAt env_at) {
    st.any_ap |= (EP != 0);
    st.any_am |= (EM != 0);
    if (EP == 0 && (st.last_ap < 0 || st.done)) return;
    if (__builtin_popcountll(EP) > 2) {                        // wave-uniform
        if (EM) pair_word_dense(st, base, EP, EM, lane, cf, out, env_at);
        else { st.last_ap = base + msb64(EP); st.done = 0; }
        return;
    }
    unsigned long long ep = EP;
    int lo = 0;                                              // the pending onset owns bits [lo, next onset)
    while (true) {
        const int p = ep ? __builtin_ctzll(ep) : 64;
        if (st.last_ap >= 0 && !st.done) {
            // (32-bit arithmetic: positions are below 2^30 and the host clamps the gate to +-2^30)
            int a = st.last_ap + cf.ival_min + 1 - base;      // first bit inside the gate
            int b = st.last_ap + cf.ival_max - base;          // one past the last bit inside the gate
            if (a < lo) a = lo;
            if (b > p) b = p;
            if (a < b) {
                const unsigned long long upto_b = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
                const unsigned long long m = EM & upto_b & ~((1ull << a) - 1ull);
                if (m) {
                    const int am = base + __builtin_ctzll(m);
                    if (st.nout < cf.cap && lane == 0) {
                        float* o = out + 3ll * st.nout;
                        o[0] = (float)st.last_ap;
                        o[1] = (float)am;
                        o[2] = env_at(am);                   // data[i, am] (:66)
                    }
                    st.nout += 1;
                    st.done = 1;
                }
            }
            if (base + 63 >= st.last_ap + cf.ival_max - 1) st.done |= (p == 64);      // gate closed
        }
        if (p == 64) break;
        st.last_ap = base + p;
        st.done = 0;
        lo = p;
        ep &= ep - 1ull;
    }
}

// Streams NR rows (NR = 1, or 2 rows whose samples arrive together) through gradient -> blur -> flags, one iteration per
// 64 samples.  Iteration c handles the gradient of sample u = 64 c + lane and the blurred gradient / flags of sample
// i = u - rad; the row has iterations 0 .. word_count(cf) - 1.
//   env_pair(u, e)   : e[r] = envelope of row r at sample u, 0 <= u < L
//   ring             : LDS, NR * ring_floats(radius) floats owned by this wave
//   taps             : LDS, 16-byte aligned, the 2 rad + 1 taps followed by zeros up to a multiple of 8 (TAPS_LDS floats)
//   [c_first, c_last]: iterations to run.  `zero_ring`: the ring holds nothing of this row yet and is zeroed (= the
//                      blur's zero padding left of the row); otherwise the call continues a stream whose previous call
//                      ended at c_first - 1.  A stream that starts inside a row must begin warm_words(rad) iterations
//                      early and ignore what the sink receives for them (the ring is being refilled).
//   sink(c, r, P, M, V, sm) : flags of iteration c, row r (P: grad > th_pos, M: grad < th_neg, V: sample is an edge
//                      candidate 0 .. L-2; ballot words) and the lane's blurred gradient (valid where in_row)
__host__ __device__ constexpr int word_count(int L, int rad) { return (L - 1 + rad) / 64 + 2; }   // incl. the flushing iteration
__host__ __device__ constexpr int warm_words(int rad) { return (2 * rad + 63) / 64; }

template <int NR, class EnvPair, class Sink>
__device__ __forceinline__ void stream_words(const Config& cf, const float* __restrict__ taps, float* __restrict__ ring, int lane,
                                             EnvPair env_pair, int c_first, int c_last, bool zero_ring, Sink sink) {
    const int L = cf.L, rad = cf.radius;
    const int RG = ring_entries(rad), rmask = RG - 1;
    const float two_sp = 2.0f * cf.spacing;
    if (zero_ring) {
        for (int q = lane; q < 2 * RG; q += 64) {
#pragma unroll
            for (int r = 0; r < NR; ++r) ring[r * 2 * RG + q] = 0.f;
        }
    }
    stof_fft::wave_lds_sync();
    // the two envelope samples behind a gradient are fetched one iteration ahead, so that rows streamed from HBM pay
    // the memory latency once and not once per 64 samples
    float ea[NR], eb[NR];
    auto fetch = [&](int u) {                                 // branch-free: clamped indices, the caller masks u >= L
        const int uc = u < L ? u : L - 1;
        const int ia = (uc >= L - 1) ? L - 1 : uc + 1;        // u = 0: 1;  u = L-1: L-1;  else u+1
        const int ib = (uc == 0) ? 0 : ((uc >= L - 1) ? L - 2 : uc - 1);
        env_pair(ia, ea);
        env_pair(ib > 0 ? ib : 0, eb);
    };
    fetch(64 * c_first + lane);
    for (int c = c_first; c <= c_last; ++c) {
        const int u = 64 * c + lane;
        float g[NR];
        const float den = (u == 0 || u == L - 1) ? cf.spacing : two_sp;
#pragma unroll
        for (int r = 0; r < NR; ++r) g[r] = (u < L && L > 1) ? (ea[r] - eb[r]) / den : 0.f;
        fetch(u + 64);
        // the ring is shared by the lanes of this wave only: wave_lds_sync() orders a lane's reads of its neighbours'
        // gradients behind their stores (and the previous word's reads ahead of these stores) for the compiler; the
        // hardware executes a wave's LDS instructions in order
        stof_fft::wave_lds_sync();
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            ring[r * 2 * RG + (u & rmask)] = g[r];
            ring[r * 2 * RG + (u & rmask) + RG] = g[r];
        }
        stof_fft::wave_lds_sync();
        // blurred gradient of sample i = u - rad from the gradients i - rad .. i + rad = u - 2 rad .. u
        const int i = u - rad;
        const bool in_row = (i >= 0) && (i < L);
        float sm[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) sm[r] = 0.f;
        // 2 rad + 1 taps, eight at a time: the taps come from LDS as two 16-byte broadcast reads (zero padded to a
        // multiple of 8 by the caller) and the ring reads of a group are in flight together; the fmaf chain keeps the
        // tap order, i.e. the reference's rounding (the padding adds zeros at the end of the chain).  Reading the taps
        // from global memory made every group wait for a vector load (vmcnt(0)): ~3x the time of this stage.
        const int ntaps = 2 * rad + 1, s0 = (u - 2 * rad) & rmask;
        auto tap_group = [&](int j) {
            const float4 ta = *reinterpret_cast<const float4*>(taps + j), tb = *reinterpret_cast<const float4*>(taps + j + 4);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float* rr = ring + r * 2 * RG + s0 + j;
                const float g0 = rr[0], g1 = rr[1], g2 = rr[2], g3 = rr[3], g4 = rr[4], g5 = rr[5], g6 = rr[6], g7 = rr[7];
                float a = sm[r];
                a = fmaf(ta.x, g0, a); a = fmaf(ta.y, g1, a); a = fmaf(ta.z, g2, a); a = fmaf(ta.w, g3, a);
                a = fmaf(tb.x, g4, a); a = fmaf(tb.y, g5, a); a = fmaf(tb.z, g6, a); a = fmaf(tb.w, g7, a);
                sm[r] = a;
            }
        };
        // straight-line code for the two radii the reference uses (rf 10: 11 taps, rf 20: 31 taps): no loop control on
        // the scalar unit, which 16 waves of a CU share
        if (ntaps <= 16) { tap_group(0); tap_group(8); }
        else if (ntaps <= 32) { tap_group(0); tap_group(8); tap_group(16); tap_group(24); }
        else for (int j = 0; j < ntaps; j += 8) tap_group(j);
        const unsigned long long V = __ballot(in_row && i < L - 1);       // an edge index is 0 .. L-2
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const unsigned long long P = __ballot(in_row && sm[r] > cf.th_pos);     // grad > thres_pos (:23)
            const unsigned long long M = __ballot(in_row && sm[r] < cf.th_neg);     // grad < thres_neg (:24)
            sink(c, r, P, M, V, in_row ? sm[r] : 0.f);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Block streamer (r3).  stream_words above is one dependent chain per 64 samples -- envelope reads -> division -> ring
// -> 2 rad + 1 fmaf -> ballots -> scalar pairing -- and a wave waits out each link: measured ~2,500 cycles per word and
// wave with 4 waves per SIMD, where its ~135 instructions need ~600.  Here an iteration takes WPI words at once
// (independent chains the hardware overlaps), the division by the wave's constant 2 g is a multiply and two FMAs, and the
// pairing leaves the loop: the flag words of up to 64 iterations collect in vector registers (FlagBatch) and are paired
// together (pair_batch below).
// ----------------------------------------------------------------------------------------------------------------
constexpr int WPI = 4;                           // words of 64 samples per iteration
// Gradient buffer of a wave (LDS): the last `hist` gradients of the previous iteration, then the 64 WPI new ones -- a
// linear image, so every blur read is `base + immediate` with no wrap arithmetic and every gradient is stored once.
__host__ __device__ constexpr int block_hist(int rad) { return rad <= 32 ? 64 : 192; }             // >= 2 rad, whole words
// (+ 8 zeros: the taps are read eight at a time, and the zero taps past 2 rad + 1 must meet finite values)
__host__ __device__ constexpr int block_buf_floats(int rad) { return block_hist(rad) + 64 * WPI + 8; }
__host__ __device__ constexpr int block_iterations(int L, int rad) { return (word_count(L, rad) + WPI - 1) / WPI; }

// RN(x / den) for den = 2 g with rcp = RN(1 / den): q = RN(x rcp) is within an ulp of the quotient, r = x - q den is exact
// in an FMA, and RN(q + r rcp) is the correctly rounded quotient (Markstein's final division step; it needs rcp correctly
// rounded -- a true division -- and no underflow in r).  Safe for |x| in [2^-63, 2^63) and for x == 0; the caller falls
// back to the true division for an iteration that holds anything else (div_unsafe_any).
__device__ __forceinline__ float div_by_const(float x, float den, float rcp) {
    const float q = x * rcp;
    const float r = __builtin_fmaf(-q, den, x);
    return __builtin_fmaf(r, rcp, q);
}
// |x| as ordered integers: largest of the four, and smallest with 0 mapped to the top (a - 1 wraps)
__device__ __forceinline__ bool div_unsafe_any(const float (&x)[4]) {
    unsigned a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = __float_as_uint(x[k]) << 1;            // drops the sign: 2 |x| bits
    const unsigned hi = max(max(a[0], a[1]), max(a[2], a[3]));
    const unsigned lo = min(min(a[0] - 1u, a[1] - 1u), min(a[2] - 1u, a[3] - 1u));
    return hi >= (0x5f000000u << 1) || lo < (0x20000000u << 1) - 1u;           // 2^63 and 2^-63
}

// Envelope accessors of the block streamer:
//   at(i)    : envelope[i], 0 <= i < L
//   ptr(i)   : address of envelope[i]; consecutive samples are STRIDE floats apart (the inner iterations read
//              ptr(u)[STRIDE * constant], which becomes an immediate offset of the load)
struct EnvRow {                                  // a row in global memory
    const float* e;
    static constexpr int STRIDE = 1;
    __device__ __forceinline__ float at(int i) const { return e[i]; }
    __device__ __forceinline__ const float* ptr(int i) const { return e + i; }
};
struct EnvPairLds {                              // row r of a pair interleaved in LDS: E[2 u + r]
    const float* e;                              // = E + r
    static constexpr int STRIDE = 2;
    __device__ __forceinline__ float at(int i) const { return e[2 * i]; }
    __device__ __forceinline__ const float* ptr(int i) const { return e + 2 * i; }
};

//   env                    : EnvRow / EnvPairLds
//   ring                   : LDS, block_buf_floats(rad) floats owned by this wave
//   sink(c, P, M, sm)      : iteration c = words c .. c + WPI - 1 (c a multiple of WPI): P[k], M[k] the flag words of word
//                            c + k (bit l <=> sample 64 (c + k) + l - rad; zero outside the row), sm[k] this lane's
//                            blurred gradient there (0 outside the row).  Words >= word_count(L, rad) are all zero.
//   batch_end(next)        : after every iteration, next = c + WPI
// The kernels are bound by the number of vector instructions they issue (4 waves per SIMD, every lane busy), so an
// iteration whose 256 samples lie inside the row -- all but the first and the last one or two -- runs a form without
// index clamps, end-of-row masks and one-sided differences (wave-uniform choice).
//   taps_g                 : the taps in global memory (optional): rad 15 (rf 20) then takes the blocked blur below, whose
//                            taps are scalar operands
//   [it_begin, it_end)     : iterations to deliver (default: the whole row).  A stream that starts inside the row runs
//                            iteration it_begin - 1 first to fill the gradient history (block_hist <= 256) and delivers
//                            nothing for it.
template <class Env, class Sink, class BatchEnd>
__device__ __forceinline__ void stream_blocks(const Config& cf, const float* __restrict__ taps, float* __restrict__ ring, int lane,
                                              const Env env, Sink sink, BatchEnd batch_end, const float* __restrict__ taps_g = nullptr,
                                              const int it_begin = 0, const int it_end = -1) {
    static_assert(WPI == 4, "div_unsafe_any and the history writes assume four words per iteration");
#if defined(__HIP_DEVICE_COMPILE__)
    // The clamped sample indices of the first and last iterations depend on the lane and the row length only; left to
    // itself the compiler computes them once per kernel and parks ~50 registers on them across the row loop (one wave
    // per SIMD less).  An opaque copy of the lane ties them to this call.
    asm volatile("" : "+v"(lane));
#endif
    const int L = cf.L, rad = cf.radius, niter = block_iterations(L, rad);
    const float sp = cf.spacing, two_sp = 2.0f * cf.spacing, rcp2 = 1.0f / two_sp;
    const int H = block_hist(rad), ntaps = 2 * rad + 1;
    for (int q = lane; q < H + 64 * WPI + 8; q += 64) ring[q] = 0.f;  // = the blur's zero padding left of the row
    // Iteration `it` covers the samples u = 256 it .. 256 it + 255 (gradients) and i = u - rad (blurred values, flags).
    // Iterations 1 .. n_in have every u in 1 .. L - 2 (central differences, no clamps) and every i in 0 .. L - 1.
    const int n_in = L >= 513 ? (L - 257) / 256 : 0;
    const int it_stop = (it_end < 0 || it_end > niter) ? niter : it_end, it0 = it_begin > 0 ? it_begin - 1 : 0;
    float ea[WPI], eb[WPI];
    float g[WPI] = {0.f, 0.f, 0.f, 0.f};
    const float* const r0 = ring + (H - 2 * rad + lane);
    // one iteration; INNER: `it` is in 1 .. n_in, NEXT_INNER: so is it + 1 (the envelope samples of the next iteration
    // are requested while this one is blurred)
    auto iteration = [&](const int it, auto inner_tag, auto next_inner_tag) {
        constexpr bool INNER = decltype(inner_tag)::value, NEXT_INNER = decltype(next_inner_tag)::value;
        const int c = WPI * it;
        // the newest H gradients (previous iteration, still in registers) become the history of this one; wave_lds_sync()
        // orders these stores behind the previous iteration's reads for the compiler -- the buffer is shared by the lanes
        // of this wave only, and the hardware executes a wave's LDS instructions in order
        stof_fft::wave_lds_sync();
        if (H == 64) ring[lane] = g[3];
        else { ring[lane] = g[1]; ring[64 + lane] = g[2]; ring[128 + lane] = g[3]; }
#pragma unroll
        for (int k = 0; k < WPI; ++k) g[k] = ea[k] - eb[k];
        if (INNER && !__ballot(div_unsafe_any(g))) {
#pragma unroll
            for (int k = 0; k < WPI; ++k) g[k] = div_by_const(g[k], two_sp, rcp2);
        } else {
#pragma unroll
            for (int k = 0; k < WPI; ++k) {
                const int u = 64 * (c + k) + lane;
                const float den = (u == 0 || u == L - 1) ? sp : two_sp;
                g[k] = (u < L && L > 1) ? g[k] / den : 0.f;
            }
        }
        if constexpr (NEXT_INNER) {
            const float* const pe = env.ptr(256 * it + 255 + lane);
#pragma unroll
            for (int k = 0; k < WPI; ++k) { ea[k] = pe[Env::STRIDE * (64 * k + 2)]; eb[k] = pe[Env::STRIDE * 64 * k]; }
        } else if (it + 1 < it_stop) {                         // clamped indices; the gradient masks u >= L
#pragma unroll
            for (int k = 0; k < WPI; ++k) {
                const int u = 256 * (it + 1) + 64 * k + lane;
                const int uc = u < L ? u : L - 1;
                ea[k] = env.at(uc + 1 < L ? uc + 1 : L - 1);   // u = 0: 1;  u = L-1: L-1;  else u+1
                eb[k] = env.at(uc > 0 ? uc - 1 : 0);           // u = 0: 0;  u = L-1: L-2;  else u-1
            }
        }
#pragma unroll
        for (int k = 0; k < WPI; ++k) ring[H + 64 * k + lane] = g[k];
        stof_fft::wave_lds_sync();                             // a lane's reads of its neighbours' gradients follow their stores
        float sm[WPI] = {0.f, 0.f, 0.f, 0.f};
        unsigned long long P[WPI] = {0ull, 0ull, 0ull, 0ull}, M[WPI] = {0ull, 0ull, 0ull, 0ull};
        if (INNER || 256 * it - rad < L) {                     // (an iteration past the row only flushes the pairing)
            // Blurred gradient of sample i = u - rad from the gradients u - 2 rad .. u; the fmaf chain keeps the tap order,
            // i.e. the reference's rounding.
            // Blocked form (rad 15): lane l takes the four consecutive outputs 4 l .. 4 l + 3 of the iteration; their
            // 2 rad + 4 gradients arrive as NV aligned 16-byte reads -- 9 floats of LDS traffic per output where the
            // word form below reads 2 rad + 1 (the blur of rf 20 was LDS-bandwidth bound: 35 of the row kernel's 53 us on
            // [4096, 4000]) -- and the taps are scalar operands.  The four results go back through the (now dead) buffer
            // to reach the word layout the flags need.
            auto blur_blocked = [&](auto rad_c) {
                constexpr int R = decltype(rad_c)::value, B = 64 - 2 * R, S = B & 3, BA = B - S, NV = (S + 4 + 2 * R + 3) / 4;
                float w[4 * NV];
                const float4* const src = reinterpret_cast<const float4*>(ring + BA + 4 * lane);
#pragma unroll
                for (int m = 0; m < NV; ++m) {
                    const float4 v = src[m];
                    w[4 * m] = v.x; w[4 * m + 1] = v.y; w[4 * m + 2] = v.z; w[4 * m + 3] = v.w;
                }
                float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j <= 2 * R; ++j) {
                    const float t = taps_g[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaf(t, w[S + e + j], a[e]);
                }
                stof_fft::wave_lds_sync();                     // every lane has its window
                *reinterpret_cast<float4*>(ring + 4 * lane) = make_float4(a[0], a[1], a[2], a[3]);
                stof_fft::wave_lds_sync();
#pragma unroll
                for (int k = 0; k < WPI; ++k) sm[k] = ring[64 * k + lane];
            };
            // Word form (any radius): lane l takes output 64 k + l of each word k, four or eight taps at a time for the WPI
            // words (tap reads from the LDS image are shared by them).
            auto tap_group4 = [&](int j) {
                const float4 ta = *reinterpret_cast<const float4*>(taps + j);
#pragma unroll
                for (int k = 0; k < WPI; ++k) {
                    const float* rr = r0 + 64 * k + j;
                    const float g0 = rr[0], g1 = rr[1], g2 = rr[2], g3 = rr[3];
                    float a = sm[k];
                    a = fmaf(ta.x, g0, a); a = fmaf(ta.y, g1, a); a = fmaf(ta.z, g2, a); a = fmaf(ta.w, g3, a);
                    sm[k] = a;
                }
                __builtin_amdgcn_sched_barrier(0);             // (keeps the scheduler from requesting every group at once: registers)
            };
            auto tap_group = [&](int j) { tap_group4(j); tap_group4(j + 4); };
            // (measured: rad 15 -- [512, 30720] split kernel 68.6 -> 61.8 us, [4096, 4000] row kernel 52.4 -> 51.1; for rad 5
            // the word form's 12 reads are no dearer and the fused kernels lose to the extra registers: 35.8 -> 48.3 us)
            if (taps_g != nullptr && rad == 15) blur_blocked(std::integral_constant<int, 15>{});
            else if (ntaps <= 12) { tap_group(0); tap_group4(8); }
            else if (ntaps <= 32) { tap_group(0); tap_group(8); tap_group(16); tap_group(24); }
            else for (int j = 0; j < ntaps; j += 8) tap_group(j);
            if constexpr (INNER) {
#pragma unroll
                for (int k = 0; k < WPI; ++k) {
                    P[k] = __ballot(sm[k] > cf.th_pos);        // grad > thres_pos (:23)
                    M[k] = __ballot(sm[k] < cf.th_neg);        // grad < thres_neg (:24)
                }
            } else {
#pragma unroll
                for (int k = 0; k < WPI; ++k) {
                    const int i = 64 * (c + k) + lane - rad;
                    const bool in_row = (i >= 0) && (i < L);
                    P[k] = __ballot(in_row && sm[k] > cf.th_pos);
                    M[k] = __ballot(in_row && sm[k] < cf.th_neg);
                    sm[k] = in_row ? sm[k] : 0.f;
                }
            }
        }
        if (it >= it_begin) sink(c, P, M, sm);
    };
    using T = std::true_type;
    using F = std::false_type;
    {                                                          // samples of the first iteration
#pragma unroll
        for (int k = 0; k < WPI; ++k) {
            const int u = 256 * it0 + 64 * k + lane;
            const int uc = u < L ? u : L - 1;
            ea[k] = env.at(uc + 1 < L ? uc + 1 : L - 1);
            eb[k] = env.at(uc > 0 ? uc - 1 : 0);
        }
    }
    for (int it = it0; it < it_stop; ++it) {
        if (it >= 1 && it < n_in && it + 1 < it_stop) iteration(it, T{}, T{});
        else if (it >= 1 && it <= n_in) iteration(it, T{}, F{});
        else iteration(it, F{}, F{});
        if (it >= it_begin) batch_end(WPI * (it + 1));         // (one call site: the pairing behind it is large)
    }
}

// 64 iterations of the pairing at a time: lane l holds the edge words (EP, EM) of the word that iteration c0 + l pairs
// (= word c0 + l - 1, samples 64 (c0 + l - 1) - rad ...); a ballot tells which iterations hold an onset, and the scalar
// loop visits only those and the stretch behind a pending onset.
template <class EnvAt>
__device__ __forceinline__ void pair_lane_words(RowState& st, unsigned long long EP, unsigned long long EM, int c0, int lane,
                                                const Config& cf, float* __restrict__ out, EnvAt env_at) {
    const int rad = cf.radius;
    {
        const unsigned long long has_ep = __ballot(EP != 0), has_em = __ballot(EM != 0);
        st.any_ap |= (has_ep != 0);
        st.any_am |= (has_em != 0);
        const unsigned ep_lo = (unsigned)EP, ep_hi = (unsigned)(EP >> 32), em_lo = (unsigned)EM, em_hi = (unsigned)(EM >> 32);
        int l = 0;
        while (l < 64) {
            if (st.last_ap < 0 || st.done) {                  // nothing pending: jump to the next iteration with an onset
                const unsigned long long rest = l ? (has_ep >> l) << l : has_ep;
                if (!rest) break;
                l = __builtin_ctzll(rest);
            }
            // (the builtin returns int: without the casts a set bit 31 of the low half would sign-extend over the high half)
            const unsigned long long ep = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ep_hi, l) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readlane(ep_lo, l);
            const unsigned long long em = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(em_hi, l) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readlane(em_lo, l);
            pair_word(st, 64 * (c0 + l - 1) - rad, ep, em, lane, cf, out, env_at);
            ++l;
        }
    }
}

// edge-candidate mask of iteration c: bit l <=> sample i = 64 c + l - rad is an edge index, 0 <= i <= L - 2
__device__ __forceinline__ unsigned long long valid_word(int c, int L, int rad) {
    int lo = rad - 64 * c, hi = L - 2 + rad - 64 * c;
    lo = lo < 0 ? 0 : lo;
    hi = hi > 63 ? 63 : hi;
    if (hi < lo) return 0ull;
    return (~0ull >> (63 - hi)) & (~0ull << lo);
}

// Flag words of up to 64 consecutive iterations kept in vector registers: lane l holds (P, M) of iteration c0 + l.
struct FlagBatch {
    unsigned plo = 0, phi = 0, mlo = 0, mhi = 0;
    unsigned long long carry_p = 0, carry_m = 0;               // (P, M) of iteration c0 - 1
    __device__ __forceinline__ void put(int lane, int idx, unsigned long long P, unsigned long long M) {
        const bool mine = lane == idx;                         // (v_writelane_b32 takes one scalar operand on this target)
        plo = mine ? (unsigned)P : plo;
        phi = mine ? (unsigned)(P >> 32) : phi;
        mlo = mine ? (unsigned)M : mlo;
        mhi = mine ? (unsigned)(M >> 32) : mhi;
    }
};

// pairs the iterations c0 .. c0 + count - 1 held by `fb` (count <= 64) and moves the carry on
template <class EnvAt>
__device__ __forceinline__ void pair_batch(RowState& st, FlagBatch& fb, int c0, int count, int lane, const Config& cf,
                                           float* __restrict__ out, EnvAt env_at) {
    const unsigned long long P = ((unsigned long long)fb.phi << 32) | fb.plo, M = ((unsigned long long)fb.mhi << 32) | fb.mlo;
    const unsigned pp_lo = __shfl_up(fb.plo, 1), pp_hi = __shfl_up(fb.phi, 1), mp_lo = __shfl_up(fb.mlo, 1), mp_hi = __shfl_up(fb.mhi, 1);
    const unsigned long long Pp = lane ? (((unsigned long long)pp_hi << 32) | pp_lo) : fb.carry_p;
    const unsigned long long Mp = lane ? (((unsigned long long)mp_hi << 32) | mp_lo) : fb.carry_m;
    const int c = c0 + lane;
    unsigned long long EP = 0, EM = 0;
    if (c >= 1 && lane < count) {                              // an edge at the last lane of a word needs the next word's first flag
        const unsigned long long Vp = valid_word(c - 1, cf.L, cf.radius);
        EP = ~Pp & ((Pp >> 1) | (P << 63)) & Vp;
        EM = ~Mp & ((Mp >> 1) | (M << 63)) & Vp;
    }
    pair_lane_words(st, EP, EM, c0, lane, cf, out, env_at);
    fb.carry_p = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(fb.phi, 63) << 32) | (unsigned)__builtin_amdgcn_readlane(fb.plo, 63);
    fb.carry_m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(fb.mhi, 63) << 32) | (unsigned)__builtin_amdgcn_readlane(fb.mlo, 63);
}

// grad_peak_detect of one row with the block streamer: flags into a FlagBatch, pairing every 64 words
template <class Env>
__device__ __forceinline__ void detect_row_blocks(const Config& cf, const float* __restrict__ taps, float* __restrict__ ring, int lane,
                                                  const Env env, float* __restrict__ out, RowState& st,
                                                  const float* __restrict__ taps_g = nullptr) {
    const int nwords = word_count(cf.L, cf.radius);
    FlagBatch fb;
    auto env_at = [&](int i) { return env.at(i); };            // amplitude of a kept peak
    stream_blocks(cf, taps, ring, lane, env,
                  [&](int c, const unsigned long long (&P)[WPI], const unsigned long long (&M)[WPI], const float (&)[WPI]) {
#pragma unroll
                      for (int k = 0; k < WPI; ++k) fb.put(lane, (c + k) & 63, P[k], M[k]);
                  },
                  [&](int next) {                              // batches start at multiples of 64 (WPI divides 64)
                      if ((next & 63) == 0 || next >= nwords) {
                          const int c0 = (next - 1) & ~63;
                          const int count = (next < nwords ? next : nwords) - c0;
                          if (count > 0) pair_batch(st, fb, c0, count, lane, cf, out, env_at);
                      }
                  },
                  taps_g);
}

// sums of the blurred gradient of one row (the Q7 pre-pass); lanes outside the row deliver 0
template <class Env>
__device__ __forceinline__ void moments_row_blocks(const Config& cf, const float* __restrict__ taps, float* __restrict__ ring, int lane,
                                                   const Env env, double (&mom)[2], const float* __restrict__ taps_g = nullptr) {
    stream_blocks(cf, taps, ring, lane, env,
                  [&](int, const unsigned long long (&)[WPI], const unsigned long long (&)[WPI], const float (&sm)[WPI]) {
#pragma unroll
                      for (int k = 0; k < WPI; ++k) {
                          mom[0] += (double)sm[k];
                          mom[1] += (double)sm[k] * (double)sm[k];
                      }
                  },
                  [](int) {}, taps_g);
}

// Pairing of a row from its stored flag words F[3 c + {0, 1, 2}] = (P, M, V) of iteration c, c = 0 .. nwords - 1 (LDS, by
// one wave), 64 iterations at a time.
template <class EnvAt>
__device__ __forceinline__ void pair_stored_words(RowState& st, const unsigned long long* __restrict__ F, int nwords, int lane,
                                                  const Config& cf, float* __restrict__ out, EnvAt env_at) {
    for (int c0 = 1; c0 < nwords; c0 += 64) {
        const int c = c0 + lane;
        unsigned long long EP = 0, EM = 0;
        if (c < nwords) {
            const unsigned long long Pp = F[3 * (c - 1)], Mp = F[3 * (c - 1) + 1], Vp = F[3 * (c - 1) + 2];
            const unsigned long long P = F[3 * c], M = F[3 * c + 1];
            EP = ~Pp & ((Pp >> 1) | (P << 63)) & Vp;
            EM = ~Mp & ((Mp >> 1) | (M << 63)) & Vp;
        }
        pair_lane_words(st, EP, EM, c0, lane, cf, out, env_at);
    }
}

// (not inlined: it runs once per row and would otherwise cost the streaming loop ~20 VGPRs = one wave per SIMD)
// echo_max reduction of one row (models/gradpeak.py:107-114 when the batch-wide echo count exceeds echo_max): keep the
// echo_max largest amplitudes, then ascending peak time -- with the reference's zero padding taking part: padded
// entries have amplitude 0 and peak 0, so a row with fewer than echo_max echoes gets its zeros in FRONT.
// `src` holds the row's min(nout, cap) echoes as written by this wave; they are re-read behind an L1-bypassing load.
__device__ __attribute__((noinline)) void reduce_row(const float* src, long long cnt, long long k, float* __restrict__ dst, int lane) {
    auto ld = [&](long long e, int f) {
        const int bits = __hip_atomic_load(reinterpret_cast<const int*>(src) + 3 * e + f, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        return __int_as_float(bits);
    };
    const long long npad = k > cnt ? k - cnt : 0;
    for (long long p = lane; p < npad; p += 64) { dst[3 * p] = 0.f; dst[3 * p + 1] = 0.f; dst[3 * p + 2] = 0.f; }
    if (cnt <= k) {
        for (long long e = lane; e < cnt; e += 64)
            for (int f = 0; f < 3; ++f) dst[3 * (npad + e) + f] = ld(e, f);
        return;
    }
    // cnt > k: k rounds of "largest amplitude not yet taken" (ties: the earlier echo), lane owns entries lane + 64 t
    const int nt = (int)((cnt + 63) / 64);
    unsigned long long taken = 0;                             // bit t: entry lane + 64 t is selected (cnt <= 4096)
    for (long long round = 0; round < k; ++round) {
        float best = -1.f;
        int best_e = 0x7fffffff;
        for (int t = 0; t < nt && t < 64; ++t) {
            const long long e = lane + 64ll * t;
            if (e < cnt && !((taken >> t) & 1ull)) {
                const float a = ld(e, 2);
                if (a > best) { best = a; best_e = (int)e; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oe = __shfl_xor(best_e, o);
            if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
        }
        if ((best_e & 63) == lane && best_e != 0x7fffffff) taken |= 1ull << (best_e >> 6);
    }
    long long pos = 0;
    for (int t = 0; t < nt && t < 64; ++t) {
        const bool sel = (taken >> t) & 1ull;
        const unsigned long long sm = __ballot(sel);
        const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (sel) {
            const long long e = lane + 64ll * t, p = pos + __builtin_popcountll(sm & lt_mask);
            for (int f = 0; f < 3; ++f) dst[3 * p + f] = ld(e, f);
        }
        pos += __builtin_popcountll(sm);
    }
}

// end of a row: counts, batch flags (Q9, Kmax) and the optional reduction
// defer_kmax: the caller folds st.nout into flags[1] itself (one atomic per work-group at the end of the kernel)
__device__ __forceinline__ void finish_row(const RowState& st, const Config& cf, long long row, float* __restrict__ out,
                                           float* __restrict__ reduced, int* __restrict__ counts, int* __restrict__ flags,
                                           int lane, bool defer_kmax = false) {
    // zero padding of the row up to `cap` (the reference pads with [0, 0, 0], :66)
    for (long long q = 3ll * (st.nout < cf.cap ? st.nout : cf.cap) + lane; q < 3 * cf.cap; q += 64) out[q] = 0.f;
    if (lane == 0) {
        counts[row] = st.nout;
        if (st.any_ap && st.any_am && st.nout == 0) atomicOr(&flags[0], 1);       // Q9 (:54-55)
        // Kmax of the batch.  One atomic per row on ONE word serialises at ~90 per microsecond (4096 rows = the whole
        // kernel); the word only grows, so a relaxed read first lets all but the first few record-setting rows skip it.
        if (!defer_kmax && st.nout > 0 && st.nout > __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&flags[1], st.nout);
    }
    if (reduced != nullptr && cf.echo_max > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                                            // this wave's echo stores have reached L2
        const long long cnt = st.nout < cf.cap ? st.nout : cf.cap;
        reduce_row(out, cnt, cf.echo_max, reduced + row * cf.echo_max * 3, lane);
    }
}

}  // namespace stof_gp
