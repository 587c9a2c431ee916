// Body sweep, round-4 form: the split-fp16 k7 layers run TWO-PASS TILE-MAJOR (included by convstack.hip inside its
// anonymous namespace; shares BodyParams / BodyLds / the packed blob with body_sweep_kernel, whose schedule -- rings, lags,
// in-place residual stream, one barrier per layer -- is unchanged: oracle/sweep_emulator.py still describes it).
//
// What changed and why (tools/micro/sweep_waves.hip, tools/micro/gen_issue_cost.py, profiles/r04_*.jsonl):
//  * one wave per SIMD issues a v_mfma_f32_16x16x32_f16 every 16.5 cycles and can slide TWO 4-cycle instructions (VALU,
//    ds_read_b128) behind each for nothing; a third costs its full issue time.  The r3 kernel ran a layer's 14 chunks
//    chunk-major and hid the epilogue (accumulator read, residual add or leaky ReLU, fp16 split, ring store: ~46 VALU per
//    16-row tile) behind the 12 MFMAs of a tile's last two chunks: 4 per MFMA, so the tail ran at ~24 cycles per MFMA and the
//    last tile's epilogue plus the layer set-up stayed exposed (3.8 k + 1.0 k of a layer's 13.6 k cycles).
//  * here a wave keeps HALF a layer's weights resident (7 chunks x 4 fragments = 112 registers per half, two halves
//    double-buffered) and sweeps its six 16-row N-tiles twice: pass A = chunks 0..6 of every tile, pass B = chunks 7..13
//    with the epilogue of tile n-1 cut into seven pieces behind tile n's 42 MFMAs (<= 8 VALU per 6 MFMAs).  Per accumulator
//    the summation order (chunk 0..13; hi*hi, hi*lo, lo*hi) is the r3 order, so every output bit is unchanged.
//  * the other half's 28 fragments are requested a whole pass (~4 k cycles) before their first use, one chunk per tile, so
//    the L1 (64 B/clk/CU; a layer's fragments are 224 KiB per CU) never sees a burst -- a fully weight-resident tile-major
//    layer has to reload all 56 fragments inside its last tile and was slower than the r3 form.
//  * the kinds of a step's layers alternate (leaky ReLU, residual add), so the step runs them as straight-line PAIRS: a branch
//    between two 500-MFMA bodies that both redefine 224 weight registers made the compiler reconcile them with ~120 moves
//    and a vmcnt(0) at every layer end.
// LDS accesses of the hand-placed stream by 32-bit LDS address: through generic pointers hipcc did the ring-row arithmetic
// in 64 bits (v_mad_u64_u32 + moves) once the values were pinned.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4_t lds_cu4_t;
typedef __attribute__((address_space(3))) u32x4_t lds_u4_t;
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ uint4 lds_ldq(unsigned a) {
    const u32x4_t v = *(lds_cu4_t*)(a);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_stq(unsigned a, uint4 v) {
    const u32x4_t w = {v.x, v.y, v.z, v.w};
    *(lds_u4_t*)(a) = w;
}

// HL (training sweeps, r4): the dumps are written as SPLIT ROWS -- [64 x fp16 hi | 64 x fp16 lo], the 256 bytes of an fp32 row, in
// the layout the rings and the weight-gradient kernel's LDS tiles use (the epilogue has both halves anyway) -- except the forward
// sweep's tensor 11 (conv12's output, read by fp32 kernels); the backward sweep then reads the saved activations' hi halves.
template <int S, int RING, int RAWRING, bool DUMP = false, bool BWD = false, bool HL = false>
__global__ __launch_bounds__(256, 1) void body_sweep_p2_kernel(const BodyParams p) {
    constexpr int PREC = STOF_PREC_F16X3;
    constexpr int NCHUNK_STEP = 11 * BODY_CHUNKS_K7;          // k7 weight chunks of a sweep step (conv_last's follow them in the blob)
    static_assert((RING & (RING - 1)) == 0 && (RAWRING & (RAWRING - 1)) == 0, "rings are powers of two");
    static_assert(S % 64 == 0 && S + 36 <= RING && S + 42 <= RAWRING, "ring must hold the live span");
    constexpr int RF = ROWF16;
    constexpr int ROWB = RF * 4;
    using Lds = BodyLds<S, RING, RAWRING, RF>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const Xr = reinterpret_cast<char*>(smem + Lds::X);
    char* const Yr = reinterpret_cast<char*>(smem + Lds::Y);
    float* const rawr = smem + Lds::RAW;
    float* const biasl = smem + Lds::BIAS;
    float* const sgl = smem + Lds::SGL;
    float* const c1l = smem + Lds::TOTAL;         // [64][10] conv1 taps + bias (BODY_P2_C1F floats behind the r3 layout)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave & 1, ni = wave >> 1;

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
    auto vmap = [&](int nv, int tl, int& n, int& tt) {
        n = nv >> p.nseg_log2;
        tt = (nv & seg_mask) * p.seg_len - p.halo + tl;
    };

    // ---- one-time setup: zero rings, biases to LDS, conv1 taps to registers
    for (int i = tid; i < Lds::RAW + RAWRING; i += 256) smem[i] = 0.f;
    for (int i = tid; i < 13 * 64; i += 256) biasl[i] = p.bias[i];
    for (int i = tid; i < BODY_P2_C1F; i += 256) c1l[i] = p.c1[i];
    const int cq = tid & 15, rl = tid >> 4;
    __syncthreads();
    // (waveforms of at least 2 S rows wrap at most once per offset: selects instead of loops -- the general loops compiled to
    // ~40 scalar instructions with a division per layer set-up)
    constexpr bool long_wf = true;                // Lp >= 2 S: the host sends shorter waveforms to the r3 kernel (body_p2_rows_ok)
    auto decode_row = [&](int nB, int tB, int off, int& n, int& t) {
        n = nB;
        t = tB + off;
        if (long_wf) {
            const bool wrap = t >= Lp;
            t -= wrap ? Lp : 0;
            n += wrap ? 1 : 0;
        } else {
            while (t >= Lp) { t -= Lp; n += 1; }
        }
    };
    // row tB - back (0 <= back <= 36) of waveform nB
    auto step_back = [&](int nB, int tB, int back, int& n, int& t) {
        n = nB;
        t = tB - back;
        if (long_wf) {
            const bool wrap = t < 0;
            t += wrap ? Lp : 0;
            n -= wrap ? 1 : 0;
        } else {
            while (t < 0) { t += Lp; n -= 1; }
        }
    };

    // relu(conv1(x)) + SemiGlobalBlock contribution for stream rows [rstart, rstart+S) -> ring dst (as body_sweep_kernel)
    const float one_x0 = opaque_one();
    auto x0_pass = [&](char* dst, int rstart, int nR, int tR, bool dump_it) {
        constexpr int NIT = S / 16;
        // the thread's 4 channels x (9 taps + bias) from LDS, ten 16-byte reads: kept in registers they cost 40 of them for the
        // whole step, and the hand-placed layers need the room (a spill inside a layer costs more than a step's worth of reloads)
        float w1[4][9], b1[4];
        {
            float wv[40];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const float4 t4 = ld4(c1l + 40 * cq + 4 * i);
                wv[4 * i] = t4.x; wv[4 * i + 1] = t4.y; wv[4 * i + 2] = t4.z; wv[4 * i + 3] = t4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int d = 0; d < 9; ++d) w1[i][d] = wv[10 * i + d];
                b1[i] = wv[10 * i + 9];
            }
        }
        float* const dump0 = (DUMP && dump_it) ? p.dump : nullptr;
        const int g0 = rstart + rl * NIT;
        if (!seg_mode && rstart >= 0 && rstart + S <= gend && tR + S <= L &&
            (p.sgb == nullptr || (tR >= p.rem_half && tR + S - p.rem_half <= SGB_SCALE * p.P))) {
            float xs[NIT + 8];
#pragma unroll
            for (int i = 0; i < NIT + 8; ++i) xs[i] = rawr[(g0 - 4 + i) & (RAWRING - 1)];
            float4 sgA = make_float4(0.f, 0.f, 0.f, 0.f), sgB = sgA;
            int sw = NIT;
            if (p.sgb != nullptr) {
                const int pos0 = tR + rl * NIT - p.rem_half;
                const int w0 = pos0 / SGB_SCALE;
                sw = SGB_SCALE * (w0 + 1) - pos0;
                const int wid = (n0 + nR) * p.P + w0;
                sgA = ld4(sgl + (wid & 7) * NF + 4 * cq);
                sgB = ld4(sgl + ((wid + 1) & 7) * NF + 4 * cq);
            }
            // two channels per v_pk_fma_f32 (the sample is the broadcast operand): half the FMA instructions of the pass, the same
            // fused multiply-adds in the same order per channel
            typedef float f2v __attribute__((ext_vector_type(2)));
            f2v wq[2][9], bq2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int d = 0; d < 9; ++d) wq[h][d] = f2v{w1[2 * h][d], w1[2 * h + 1][d]};
                bq2[h] = f2v{b1[2 * h], b1[2 * h + 1]};
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const bool first = it < sw;
                const float sg[4] = {first ? sgA.x : sgB.x, first ? sgA.y : sgB.y, first ? sgA.z : sgB.z, first ? sgA.w : sgB.w};
                float v[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f2v a = bq2[h];
#pragma unroll
                    for (int d = 0; d < 9; ++d) a = __builtin_elementwise_fma(wq[h][d], f2v{xs[it + d], xs[it + d]}, a);
                    v[2 * h] = fmaxf(a[0], 0.f) + sg[2 * h];
                    v[2 * h + 1] = fmaxf(a[1], 0.f) + sg[2 * h + 1];
                }
                char* const row = dst + ((g0 + it) & (RING - 1)) * ROWB;
                if (!HL && dump0 != nullptr)
                    st4(dump0 + ((size_t)(n0 + nR) * Ltrue + tR + rl * NIT + it) * NF + 4 * cq, make_float4(v[0], v[1], v[2], v[3]));
                const half2v h01 = cvt_h2(v[0], v[1]), h23 = cvt_h2(v[2], v[3]);
                half2v l01, l23;                    // v - hi is exact in fp32: one rounding to fp16 (v_fma_mixlo_f16 / mixhi)
                l01[0] = (_Float16)mix_sub(h01[0], v[0], one_x0); l01[1] = (_Float16)mix_sub(h01[1], v[1], one_x0);
                l23[0] = (_Float16)mix_sub(h23[0], v[2], one_x0); l23[1] = (_Float16)mix_sub(h23[1], v[3], one_x0);
                if (HL && dump0 != nullptr) {
                    char* const drow = reinterpret_cast<char*>(dump0 + ((size_t)(n0 + nR) * Ltrue + tR + rl * NIT + it) * NF);
                    *reinterpret_cast<uint2*>(drow + 8 * cq) = make_uint2(h2_bits(h01), h2_bits(h23));
                    *reinterpret_cast<uint2*>(drow + 128 + 8 * cq) = make_uint2(h2_bits(l01), h2_bits(l23));
                }
                *reinterpret_cast<uint2*>(row + 8 * cq) = make_uint2(h2_bits(h01), h2_bits(h23));
                *reinterpret_cast<uint2*>(row + 128 + 8 * cq) = make_uint2(h2_bits(l01), h2_bits(l23));
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
            if (t >= Lp) { t -= Lp; nl += 1; }
            int nw, tw;
            vmap(n0 + nl, t, nw, tw);
            const bool ok = (g >= 0) && (g < gend) && (t < L) && (tw >= 0) && (tw < Ltrue);
            float4 sg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.sgb != nullptr && ok) {
                const int pos = tw - p.rem_half;
                if (pos >= 0 && pos < SGB_SCALE * p.P) {
                    const int wid = nw * p.P + pos / SGB_SCALE;
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
            if (!HL && dump0 != nullptr && ok && t >= p.halo && t < p.halo + p.seg_len)
                st4(dump0 + ((size_t)nw * Ltrue + tw) * NF + 4 * cq, o);
            char* const lrow = dst + ((g0 + it) & (RING - 1)) * ROWB;
            store_act4<PREC>(lrow, 4 * cq, o);
            if (HL && dump0 != nullptr && ok && t >= p.halo && t < p.halo + p.seg_len) {      // the split image just written (same thread)
                char* const drow = reinterpret_cast<char*>(dump0 + ((size_t)nw * Ltrue + tw) * NF);
                *reinterpret_cast<uint2*>(drow + 8 * cq) = *reinterpret_cast<const uint2*>(lrow + 8 * cq);
                *reinterpret_cast<uint2*>(drow + 128 + 8 * cq) = *reinterpret_cast<const uint2*>(lrow + 128 + 8 * cq);
            }
        }
    };

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
        if (tid < 4 && tid < L) {
            int nw, tw;
            vmap(n0, tid, nw, tw);
            if (tw >= 0 && tw < Ltrue) rawr[tid] = p.x[(size_t)nw * Ltrue + tw];
        }
        fetch_step(S, 0, 0);
    }
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
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (HL) {
                // g6 arrives as split rows: the thread's 4 channels are 8 bytes of the hi part and 8 of the lo part -- the ring's own layout
                if (ok) {
                    const char* const row = reinterpret_cast<const char*>(p.gin + ((size_t)nw * Ltrue + tw) * NF);
                    const uint2 hi = *reinterpret_cast<const uint2*>(row + 8 * cq), lo = *reinterpret_cast<const uint2*>(row + 128 + 8 * cq);
                    v[it] = make_float4(__uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(lo.x), __uint_as_float(lo.y));
                }
            } else if (ok) {
                v[it] = ld4(p.gin + ((size_t)nw * Ltrue + tw) * NF + 4 * cq);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            char* const row = dst + ((g0 + it) & (RING - 1)) * ROWB;
            if constexpr (HL) {
                *reinterpret_cast<uint2*>(row + 8 * cq) = make_uint2(__float_as_uint(v[it].x), __float_as_uint(v[it].y));
                *reinterpret_cast<uint2*>(row + 128 + 8 * cq) = make_uint2(__float_as_uint(v[it].z), __float_as_uint(v[it].w));
            } else {
                store_act4<PREC>(row, 4 * cq, v[it]);
            }
        }
    };

    // ---- weights: two half-layer buffers of 7 chunks x 4 fragments (M-tile 0 hi | lo, M-tile 1 hi | lo), see the header
    constexpr int HC = BODY_CHUNKS_K7 / 2;
    // fragment f of chunk c for this wave's 32-channel block, as a raw buffer load: resource = the chunk blob (scalar registers),
    // vector offset = 16 lane, scalar offset = the fragment's byte offset -- no vector address arithmetic and one s_add per load
    // (with flat 64-bit lane addresses hipcc hoisted 28 addresses per pass out of the step loop, spilled them and reloaded each
    // behind vmcnt(0); computed in place they cost a 64-bit VALU add per load)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.chunks), 0, 0x7fffffff, 0x00020000);
    const int woff = lane * 16;
    auto wload = [&](int c, int f) -> uint4 {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, (c * FRAGS_PER_CHUNK + f) * 2048 + mi * 1024, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    uint4 W[2][HC][FRAGS_PER_CHUNK];
#pragma unroll
    for (int c = 0; c < HC; ++c)
#pragma unroll
        for (int f = 0; f < FRAGS_PER_CHUNK; ++f) W[0][c][f] = wload(c, f);

#ifdef STOF_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = stamp();
    const unsigned long long tstart = tprev;
#endif
    constexpr int NN = S / 32;                    // N-tiles (16 rows) per wave
    static_assert(NN == 6, "the two-pass layer is written for six N-tiles per wave (S = 192)");
#ifndef STOF_P2_PD
#define STOF_P2_PD 2
#endif
    constexpr int PD = STOF_P2_PD, NB = PD + 1;   // activation fragments are requested PD units ahead of their MFMAs
    const int i16 = lane & 15, q4 = lane >> 4;
    auto mfma16 = [](const uint4& a, const uint4& b, floatx4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(a), as_h8(b), c, 0, 0, 0);
    };
    const float one = opaque_one();

    const int nsteps = (gend - GAP + LAG_LAST + S - 1) / S;
    int nS = 0, tS = 0;                           // waveform / time of stream row F - S (wave-uniform)
    int F = 0;

    // One k7 layer.  KIND: 0 = residual add in place, 1 = leaky ReLU (forward), 2 = plain, 3 = times lrelu'(saved) (backward).
    //
    // The instruction stream is placed BY HAND: a unit = one (N-tile, chunk) = 6 MFMAs, and behind every MFMA sit at most two
    // single-issue instructions (free beside a 16x16x32 MFMA; a third costs its whole issue time: tools/micro/gen_issue_cost.py,
    // profiles/r04_issue_cost.jsonl), closed by a scheduling barrier so the compiler keeps them there -- left to the
    // sched_group_barrier solver the epilogue's VALU clumped in groups of 8 between MFMAs and pass B ran at 22.7 cycles per MFMA.
    //   gap 0: ds_read hi fragment of the unit PD ahead | E        gap 3: E, A
    //   gap 1: ds_read lo fragment                      | E        gap 4: E, A
    //   gap 2: E, E                                                gap 5: A, G
    // (scalar instructions and s_waitcnt take the same issue slots as vector ones: two per gap in all)
    // E = the next instruction of the previous tile's epilogue queue (pass B; 42 slots per tile for its <= 41 instructions),
    // A = the three instructions of the LDS address of the unit PD + 1 ahead when its tap changes, G = one weight-fragment load
    // for the other half-layer buffer (28 per pass).
    auto layer = [&](auto kind_c, const int j, auto hl_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr bool LHL = decltype(hl_c)::value;   // this layer's dump as split rows
        constexpr bool INPL = KIND == 0;
        int nR, tR;                               // waveform / time of the layer's first row R0 = F - S - lag (may precede the stream)
        step_back(nS, tS, 3 * j, nR, tR);
        const char* const src = (j & 1) ? Xr : Yr;             // odd sweep layers read ring X and write ring Y
        char* const dst = (j & 1) ? Yr : Xr;
        const int R0 = F - S - 3 * j;
        floatx4 bvec[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float4 bb = ld4(biasl + j * 64 + 32 * mi + 8 * q4 + 4 * m);
            bvec[m][0] = bb.x; bvec[m][1] = bb.y; bvec[m][2] = bb.z; bvec[m][3] = bb.w;
        }
        auto row_of = [&](int n, bool& valid, int& slot, int& nw, int& tw, int& tk) {
            const int off = 16 * (NN * ni + n) + i16;
            const int g = R0 + off;
            int nk;
            decode_row(nR, tR, off, nk, tk);
            vmap(n0 + nk, tk, nw, tw);
            valid = (g >= 0) && (g < gend) && (tk < L) && (tw >= 0) && (tw < Ltrue);
            slot = (g & (RING - 1)) * ROWB + (32 * mi + 8 * q4) * 2;
        };
        // wave-uniform: the wave's 96-row span inside one waveform (and, in segment mode, inside the segment's own rows)?
        const int offw0 = 16 * NN * ni, gw0 = R0 + offw0;
        int nkw, tkw, nww, tww;
        decode_row(nR, tR, offw0, nkw, tkw);
        vmap(n0 + nkw, tkw, nww, tww);
        const bool span_ok = (gw0 >= 0) && (gw0 + 16 * NN - 1 < gend) && (tkw + 16 * NN - 1 < L) && (tww >= 0) &&
                             (tww + 16 * NN - 1 < Ltrue) && (tkw >= p.halo) && (tkw + 16 * NN - 1 < p.halo + p.seg_len);
        float* const dumpj = DUMP ? p.dump + (size_t)j * p.dump_stride : nullptr;
        // the lane's two 16-byte stores of its row of tile t: fp32 rows -> channels 8 q4 .. + 3 | + 4 .. + 7 (bytes + 0, + 16);
        // split rows -> hi | lo of the 8 channels (bytes + 0, + 128).  Spans with padding rows store into a scrap area behind the
        // twelve tensors (32 bytes per lane) and the fix-up below writes the valid rows from the LDS image.
        char* dlane = nullptr;
        long long dstep = 0;
        int dsecond = 16;
        if constexpr (DUMP) {
            if (span_ok) {
                dlane = reinterpret_cast<char*>(dumpj + ((size_t)nww * Ltrue + tww + i16) * NF) + (32 * mi + 8 * q4) * (LHL ? 2 : 4);
                dstep = 16 * NF * 4;
                dsecond = LHL ? 128 : 16;
            } else {
                dlane = reinterpret_cast<char*>(p.dump + 12 * p.dump_stride + 8 * lane);
            }
        }
        // backward, masked layers: saved activation of the lane's 8 channels of its row of N-tile n (the forward dump's tensor
        // 1 + 2 k, k = 5 - j/2), requested a whole pass (~2 us) before the epilogue that uses it: behind pass-A tile n for pass-B tile n
        // (r4: requested one tile ahead, ~0.4 us, every epilogue waited out the rest of the HBM round trip -- SQ_WAIT_ANY 0.42)
        // HL: the forward dump holds split rows; the mask needs the sign of the value = the sign of its hi half (16 bytes per lane and
        // tile instead of 32).  An activation with |y| < 2^-25 has hi = +-0 and reads as 'not positive' (slope 0.01) whatever its sign --
        // the forward arithmetic saw such a value as hi + lo ~ 0 too.
        const float* const ysp = (BWD && KIND == 3) ? p.fwd_dump + (size_t)(1 + 2 * (5 - (j >> 1))) * p.dump_stride + (HL ? 16 * mi + 4 * q4 : 32 * mi + 8 * q4) : nullptr;
        float4 ysv[KIND == 3 ? NN : 1][HL ? 1 : 2];
        auto ys_load = [&](int n) {
            if constexpr (KIND == 3) {
                // ONE unconditional load per tile: rows that are not valid (gap rows, stream ends) read row 0 of the tensor -- their
                // results are overwritten by the padding fix-up below.  (A conditional load with a constant default made the compiler
                // fold the `> 0` compares into the load's block, behind an s_waitcnt vmcnt(0): every request waited out its own HBM
                // round trip, 55 k of the backward step's 200 k cycles.)
                size_t row;
                if (span_ok) {
                    row = (size_t)nww * Ltrue + tww + i16 + 16 * n;
                } else {
                    bool valid;
                    int slot, nw, tw, tk;
                    row_of(n, valid, slot, nw, tw, tk);
                    row = valid ? (size_t)nw * Ltrue + tw : 0;
                }
                const float* const s0 = ysp + row * NF;
                ysv[n][0] = ld4(s0);
                if constexpr (!HL) ysv[n][1] = ld4(s0 + 4);
            }
        };

        // ---- activation fragments: unit g = (pass ps, tile n, chunk c) reads rows rbase + 16 n + d of src, d = tap of chunk
        // cc = 7 ps + c, at byte 64 (cc & 1) (+ 128: lo) of the lane's 16-byte column.  `aq` = LDS byte address of (tile, tap).
        const unsigned lds_base = (unsigned)(size_t)(lds_char_t*)smem;
        const unsigned src_a = lds_base + ((j & 1) ? Lds::X : Lds::Y) * 4, dst_a = lds_base + ((j & 1) ? Lds::Y : Lds::X) * 4;
        const int rbase = R0 + 16 * NN * ni + i16 - 3;          // row of N-tile 0, tap 0
        const unsigned asrc = src_a + 16 * q4;
        const unsigned slotdelta = dst_a - src_a + 64 * mi;     // from the (tile, tap 3) operand address to the epilogue's slot in dst
        unsigned at = 0, aq = 0;
        auto unit_cc = [](int g) { return (g / (NN * HC)) * HC + g % HC; };
        auto unit_tile = [](int g) { return (g / HC) % NN; };
        auto new_tap = [&](int g) { return g % HC == 0 || (unit_cc(g) & 1) == 0; };
        // PIN(x): an empty volatile asm that 'rewrites' x.  Pure arithmetic floats freely in the compiler's DAG (it is linearised
        // right in front of its first user, and only then do the scheduling barriers freeze the order): the pin is a user that
        // sits where the instruction is wanted.  Never on a load's result: the pin would be a wait for it.
#define STOF_PIN(x) asm volatile("" : "+v"(x))
        auto addr_op = [&](int g, int k) {                       // instruction k of the address of unit g
            if (k == 0) { at = (unsigned)(rbase + (16 * unit_tile(g) + (unit_cc(g) >> 1))); STOF_PIN(at); }
            else if (k == 1) { at = at & (RING - 1); STOF_PIN(at); }
            else { aq = __umul24(at, ROWB) + asrc; STOF_PIN(aq); }
        };
        auto frag_read = [&](uint4& b, int g, int part) { b = lds_ldq(aq + 64 * (unit_cc(g) & 1) + 128 * part); };

        // ---- epilogue of N-tile t (8 consecutive channels of one row per lane) as a queue of single instructions
        struct Epi { uint4 oh, ol; float v[8], w[8]; unsigned h[4], l[4]; };
        floatx4 acc[NN][2];
        unsigned slotq[3];                                            // LDS byte address of the lane's 16 bytes of its row of tile t (hi image); tile t's
                                                                 // is taken two units before pass-B tile t starts and used until tile t + 1 ends
        // instruction q (0..41) of the queue that runs behind the MFMAs of pass-B tile t + 1 (or exposed, for the last tile)
        auto epi_op = [&](Epi& e, int t, int q) {
            const unsigned sl = slotq[t % 3];
            if (q < 2) {
                if constexpr (INPL) { if (q == 0) e.oh = lds_ldq(sl); else e.ol = lds_ldq(sl + 128); }
            } else if (q < 10) {
                const int x = q - 2;
                e.v[x] = acc[t][x >> 2][x & 3];
                STOF_PIN(e.v[x]);
            } else if (q < 26) {
                const int k = q - 10;
                if constexpr (INPL) {                            // + old hi (k < 8), + old lo (k >= 8): value x of pair x >> 1
                    const int x = k & 7, pr = x >> 1;
                    const uint4& o = k < 8 ? e.oh : e.ol;
                    const half2v hv = bits_h2(pr == 0 ? o.x : pr == 1 ? o.y : pr == 2 ? o.z : o.w);
                    e.v[x] = mix_add(hv[x & 1], e.v[x], one);
                    STOF_PIN(e.v[x]);
                } else if constexpr (KIND == 1) {                // leaky_relu(v, 0.01) = med3(v, 0.01 v, huge): 8 v_mul, then 8 v_med3
                    const int x = k & 7;                          // (a dependent pair back to back stalls the wave for the VALU latency)
                    if (k < 8) { e.w[x] = 0.01f * e.v[x]; STOF_PIN(e.w[x]); }
                    else { e.v[x] = __builtin_amdgcn_fmed3f(e.v[x], e.w[x], 3.0e38f); STOF_PIN(e.v[x]); }
                } else if constexpr (KIND == 3) {                // times lrelu'(saved activation): one value per slot
                    if (k < 8) {
                        if constexpr (HL) {
                            const float4 sv = ysv[KIND == 3 ? t : 0][0];      // 8 fp16 hi halves: value k = half k & 1 of word k >> 1
                            const float wf = (k >> 1) == 0 ? sv.x : (k >> 1) == 1 ? sv.y : (k >> 1) == 2 ? sv.z : sv.w;
                            const half2v hv = bits_h2(__float_as_uint(wf));
                            e.v[k] = hv[k & 1] > (_Float16)0.f ? e.v[k] : 0.01f * e.v[k];
                        } else {
                            const float4 sv = ysv[KIND == 3 ? t : 0][k >> 2];
                            const float sx = (k & 3) == 0 ? sv.x : (k & 3) == 1 ? sv.y : (k & 3) == 2 ? sv.z : sv.w;
                            e.v[k] = sx > 0.f ? e.v[k] : 0.01f * e.v[k];
                        }
                        STOF_PIN(e.v[k]);
                    }
                }
            } else if (q < 30) {
                const int pr = q - 26;
                e.h[pr] = h2_bits(cvt_h2(e.v[2 * pr], e.v[2 * pr + 1]));
                STOF_PIN(e.h[pr]);
            } else if (q == 30) {
                // (slot of the tile after next: see the call sites; nothing of this tile)
            } else if (q < 39) {
                // lo = fp16(v - hi): v_fma_mixlo_f16 / v_fma_mixhi_f16 (the difference is exact in fp32, so the one rounding equals
                // cvt(fma)); the four low halves first, then the four high halves: mixhi reads the register its mixlo wrote
                const int k = q - 31, pr = k & 3;
                if (k < 4) asm volatile("v_fma_mixlo_f16 %0, -%1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(e.l[pr]) : "v"(e.h[pr]), "s"(one), "v"(e.v[2 * pr]));
                else asm volatile("v_fma_mixhi_f16 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(e.l[pr]) : "v"(e.h[pr]), "s"(one), "v"(e.v[2 * pr + 1]));
            } else if (q == 39) {
                lds_stq(sl, make_uint4(e.h[0], e.h[1], e.h[2], e.h[3]));
            } else if (q == 40) {
                lds_stq(sl + 128, make_uint4(e.l[0], e.l[1], e.l[2], e.l[3]));
            } else if (q == 41) {
                if constexpr (DUMP) {
                    if constexpr (LHL) {
                        *reinterpret_cast<uint4*>(dlane + t * dstep) = make_uint4(e.h[0], e.h[1], e.h[2], e.h[3]);
                        *reinterpret_cast<uint4*>(dlane + t * dstep + dsecond) = make_uint4(e.l[0], e.l[1], e.l[2], e.l[3]);
                    } else {
                        st4(reinterpret_cast<float*>(dlane + t * dstep), make_float4(e.v[0], e.v[1], e.v[2], e.v[3]));
                        st4(reinterpret_cast<float*>(dlane + t * dstep + dsecond), make_float4(e.v[4], e.v[5], e.v[6], e.v[7]));
                    }
                }
            }
        };
        const int cthis = (j - 1) * BODY_CHUNKS_K7;
        const int cnext = (j * BODY_CHUNKS_K7) % NCHUNK_STEP;          // the layer after the step's last one is the next step's first
        constexpr int NU = 2 * NN * HC;                                // units per layer
        Epi ep;                                   // one is enough: a tile's queue is over before the next tile's starts
        uint4 bq[NB][2];
        // prologue: address of unit 0, fragments of units 0 .. PD-1 (same tap), address of unit PD
        static_assert(PD == 2 || PD == 3, "the prologue is written for a prefetch distance of two or three units");
#pragma unroll
        for (int k = 0; k < 3; ++k) addr_op(0, k);
        frag_read(bq[0][0], 0, 0); frag_read(bq[0][1], 0, 1);
        frag_read(bq[1][0], 1, 0); frag_read(bq[1][1], 1, 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) addr_op(2, k);
        if constexpr (PD == 3) { frag_read(bq[2][0], 2, 0); frag_read(bq[2][1], 2, 1); }   // unit 3 shares unit 2's tap
        STAMP_ADD(3);                             // layer set-up
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
#ifdef STOF_STAMP_PASSA
            if (ps == 1) STAMP_ADD(3);            // diagnostic: pass A is then counted with the layer set-up (slot 3)
#endif
#pragma unroll
            for (int n = 0; n < NN; ++n) {
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    const int g = (ps * NN + n) * HC + c;              // position in the layer's sequence of 84 units
                    const uint4 (&b)[2] = bq[g % NB];
                    const uint4 (&w)[FRAGS_PER_CHUNK] = W[ps][c];
                    floatx4 (&a)[2] = acc[n];
                    const bool first = ps == 0 && c == 0;            // the layer's bias enters as the C operand of the first MFMA
                    const bool epi = ps == 1 && n > 0;               // the previous tile's epilogue rides behind this unit
                    Epi& e = ep;
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        // hi*hi, hi*lo, lo*hi of M-tiles 0 and 1, alternating accumulators
                        a[i & 1] = mfma16(w[2 * (i & 1) + (i >= 4 ? 1 : 0)], b[(i >> 1) == 1 ? 1 : 0], (first && i < 2) ? bvec[i & 1] : a[i & 1]);
                        // E slots of the unit: gaps 0, 1 (one each, beside the fragment reads), 2 (two), 3, 4 (one each)
                        const int ga = g + PD + 1;                     // its fragments are requested in the next unit
                        const bool na = ga < NU && new_tap(ga);
                        if (i < 2) {
                            if (g + PD < NU) frag_read(bq[(g + PD) % NB][i], g + PD, i);
                            if (epi) epi_op(e, n - 1, 6 * c + i);
                            // the slot of pass-B tile t is its (tile, tap 3) operand address, live in `aq` while unit NN HC + HC t - PD runs
                            if (i == 0 && g >= NN * HC - PD && (g + PD) % HC == 0) { slotq[unit_tile(g + PD) % 3] = aq + slotdelta; STOF_PIN(slotq[unit_tile(g + PD) % 3]); }
                        } else if (i == 2) {
                            if (epi) { epi_op(e, n - 1, 6 * c + 2); epi_op(e, n - 1, 6 * c + 3); }
                        } else if (i < 5) {
                            if (epi) epi_op(e, n - 1, 6 * c + i + 1);
                            // the last tile's own old-value reads (its queue runs exposed after the loop): the previous tile's
                            // queue is past its last use of them, and these two gaps of the layer's last unit carry no address
                            if (ps == 1 && n == NN - 1 && c == HC - 1) epi_op(e, n, i - 3);
#ifndef STOF_P2_DIAG_NOA
                            if (na) addr_op(ga, i - 3);                // one address instruction per gap: the three form a dependent chain
#endif
                        } else {
#ifndef STOF_P2_DIAG_NOA
                            if (na) addr_op(ga, 2);
#endif
                            // the other half's weights a whole pass ahead of their use: one fragment per unit (28 of a pass's 42)
                            const int li = n * (HC - 1) + c;
#ifdef STOF_P2_DIAG_NOG
                            if (false) {
#else
                            if (c < HC - 1 && li < HC * FRAGS_PER_CHUNK) {
#endif
                                const int cw = li / FRAGS_PER_CHUNK, f = li % FRAGS_PER_CHUNK;
                                W[ps ^ 1][cw][f] = wload((ps == 0 ? cthis + HC : cnext) + cw, f);
                            }
                            if (BWD && KIND == 3 && ps == 0 && c == HC - 1) ys_load(n);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        STAMP_ADD(4);                             // both passes incl. the overlapped epilogues
        // the last tile's epilogue has nothing to hide behind
#pragma unroll
        for (int q = 2; q < 42; ++q) epi_op(ep, NN - 1, q);
        // Rows outside [0, L) of their waveform (gap rows, stream ends, segment padding) must read as zeros for the next layer
        // (= its zero padding): the lanes of padding rows overwrite what the branch-free epilogue stored (same lane, program
        // order: no race; the barrier follows), and the training dump takes the valid rows of the span from the LDS image.
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
                    if constexpr (LHL) {
                        char* const o = reinterpret_cast<char*>(dumpj + ((size_t)nw * Ltrue + tw) * NF) + (32 * mi + 8 * q4) * 2;
                        *reinterpret_cast<uint4*>(o) = ldq(dst + slot);
                        *reinterpret_cast<uint4*>(o + 128) = ldq(dst + slot + 128);
                        continue;
                    }
                    const half8 hh8 = as_h8(ldq(dst + slot)), ll8 = as_h8(ldq(dst + slot + 128));
                    float* const o = dumpj + ((size_t)nw * Ltrue + tw) * NF + 32 * mi + 8 * q4;
                    st4(o, make_float4((float)hh8[0] + (float)ll8[0], (float)hh8[1] + (float)ll8[1],
                                       (float)hh8[2] + (float)ll8[2], (float)hh8[3] + (float)ll8[3]));
                    st4(o + 4, make_float4((float)hh8[4] + (float)ll8[4], (float)hh8[5] + (float)ll8[5],
                                           (float)hh8[6] + (float)ll8[6], (float)hh8[7] + (float)ll8[7]));
                }
            }
        }
        STAMP_ADD(5);                             // exposed epilogue (last tile) + padding fix-up
        __syncthreads();
        STAMP_ADD(2);
#undef STOF_PIN
    };

    // conv_last with r <= 16: one 16-channel output tile on v_mfma_f32_16x16x32_f16, every wave 48 rows of the step (as r3)
    auto conv_last16 = [&]() {
        int nR, tR;
        step_back(nS, tS, LAG_LAST, nR, tR);
        const char* const src = Yr;               // sweep layer 12 reads conv12's output
        const int R0 = F - S - LAG_LAST;
        const int j16 = lane & 15;
        const uint4* lw = reinterpret_cast<const uint4*>(p.last16) + lane;
        uint4 wh[BODY_CHUNKS_LAST], wl[BODY_CHUNKS_LAST];
#pragma unroll
        for (int cc = 0; cc < BODY_CHUNKS_LAST; ++cc) { wh[cc] = lw[(cc * 2) * 64]; wl[cc] = lw[(cc * 2 + 1) * 64]; }
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
                // fused arg-max picker: per (virtual) waveform of this tile -- at most two, the tile is 16 consecutive stream
                // rows -- its max / min and the positions equal to the max
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
                    const unsigned long long mm = __ballot(mine) & 0xffffull;
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
                        uint4* d4 = reinterpret_cast<uint4*>(p.onset_ws + (size_t)nwX * p.onset_slots + slot);
                        d4[0] = make_uint4(1u, (unsigned)tw_base, __float_as_uint(m), __float_as_uint(lo));
                        d4[1] = make_uint4((unsigned)eq[0], (unsigned)(eq[0] >> 32), (unsigned)eq[1], (unsigned)(eq[1] >> 32));
                        d4[2] = make_uint4((unsigned)eq[2], (unsigned)(eq[2] >> 32), (unsigned)eq[3], (unsigned)(eq[3] >> 32));
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

    // conv_last with r > 16 (64-wide output block, of which r channels are real): tile-major on the wave's six N-tiles, its
    // 6 chunks in weight buffer 1 (free: conv12's pass B is over; buffer 0 already holds the next step's first half-layer)
    auto conv_last_wide = [&]() {
        int nR, tR;
        step_back(nS, tS, LAG_LAST, nR, tR);
        const char* const src = Yr;
        const int R0 = F - S - LAG_LAST;
#pragma unroll
        for (int cc = 0; cc < BODY_CHUNKS_LAST; ++cc)
#pragma unroll
            for (int f = 0; f < FRAGS_PER_CHUNK; ++f) W[1][cc][f] = wload(NCHUNK_STEP + cc, f);
        floatx4 bvec[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float4 bb = ld4(biasl + 12 * 64 + 32 * mi + 8 * q4 + 4 * m);
            bvec[m][0] = bb.x; bvec[m][1] = bb.y; bvec[m][2] = bb.z; bvec[m][3] = bb.w;
        }
        const int rbase = R0 + 16 * NN * ni + i16 - 1;
        bool bad = false;
#pragma unroll 1
        for (int n = 0; n < NN; ++n) {
            uint4 b[BODY_CHUNKS_LAST][2];
#pragma unroll
            for (int cc = 0; cc < BODY_CHUNKS_LAST; ++cc) {
                const char* row = src + ((rbase + 16 * n + (cc >> 1)) & (RING - 1)) * ROWB + 64 * (cc & 1) + 16 * q4;
                b[cc][0] = ldq(row);
                b[cc][1] = ldq(row + 128);
            }
            floatx4 a[2] = {bvec[0], bvec[1]};
#pragma unroll
            for (int cc = 0; cc < BODY_CHUNKS_LAST; ++cc) {
                const uint4 (&w)[FRAGS_PER_CHUNK] = W[1][cc];
                a[0] = mfma16(w[0], b[cc][0], a[0]);
                a[1] = mfma16(w[2], b[cc][0], a[1]);
                a[0] = mfma16(w[0], b[cc][1], a[0]);
                a[1] = mfma16(w[2], b[cc][1], a[1]);
                a[0] = mfma16(w[1], b[cc][0], a[0]);
                a[1] = mfma16(w[3], b[cc][0], a[1]);
            }
            const int off = 16 * (NN * ni + n) + i16;
            const int g = R0 + off;
            int nk, tk, nw, tw;
            decode_row(nR, tR, off, nk, tk);
            vmap(n0 + nk, tk, nw, tw);
            // conv_last + SampleShuffle1D: out[n][t*r + k] = conv_last[n][k][t]; only the segment's own rows
            const bool valid = (g >= 0) && (g < gend) && (tk < L) && (tw >= 0) && (tw < Ltrue) && (tk >= p.halo) && (tk < p.halo + p.seg_len);
            const float v[8] = {a[0][0], a[0][1], a[0][2], a[0][3], a[1][0], a[1][1], a[1][2], a[1][3]};
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
        STAMP_ADD(5);
        __syncthreads();
        STAMP_ADD(2);
    };

    for (int step = 1; step <= nsteps; ++step) {
        F = step * S;
        if (step > 1) {
            tS += S;
            while (tS >= Lp) { tS -= Lp; nS += 1; }
        }
        // land the raw samples [F+4-S, F+4) and the SemiGlobalBlock rows fetched one step ago, start the next step's fetch
        if constexpr (!BWD) {
            if (tid < S) rawr[(F + 4 - S + tid) & (RAWRING - 1)] = raw_next;
            if (sg_slot >= 0) sgl[sg_slot * NF + (tid & 63)] = sg_next;
            int nN = nS, tN = tS + S;
            while (tN >= Lp) { tN -= Lp; nN += 1; }
            fetch_step(F + S, nN, tN);
        }
        __syncthreads();
        STAMP_ADD(0);
        if constexpr (BWD) gin_pass(Xr, F - S, nS, tS);
        else x0_pass(Xr, F - S, nS, tS, true);    // sweep layer 0
        STAMP_ADD(1);
        __syncthreads();
        STAMP_ADD(2);
        if constexpr (!BWD) {
            // conv2 .. conv11: (leaky ReLU, residual add) x 5; then the long skip seeds ring Y with x0 and conv12 adds in place
#pragma unroll 1
            for (int pp = 0; pp < 5; ++pp) {
                layer(std::integral_constant<int, 1>{}, 2 * pp + 1, std::bool_constant<HL>{});
                layer(std::integral_constant<int, 0>{}, 2 * pp + 2, std::bool_constant<HL>{});
            }
            {
                int nR, tR;
                step_back(nS, tS, 33, nR, tR);
                x0_pass(Yr, F - S - 33, nR, tR, false);
                STAMP_ADD(1);
                __syncthreads();
                STAMP_ADD(2);
            }
            {
                // the layer number as a value the optimiser cannot see through: with a literal 11 it hoists the lane addresses of
                // the layer's 56 weight fragments out of the step loop (112 registers), spills them and reloads each behind vmcnt(0)
                int j12;
                asm volatile("s_mov_b32 %0, 11" : "=s"(j12));
                layer(std::integral_constant<int, 0>{}, j12, std::false_type{});      // conv12's output stays fp32 (conv_last's weight gradient reads it)
            }
            if (p.last16 != nullptr) conv_last16();
            else conv_last_wide();
        } else {
            // conv12^T plain, then (conv(2k+3)^T times lrelu'(saved), conv(2k+2)^T added in place) x 5
            {
                int j1;                             // opaque for the same reason as conv12's layer number in the forward sweep
                asm volatile("s_mov_b32 %0, 1" : "=s"(j1));
                layer(std::integral_constant<int, 2>{}, j1, std::bool_constant<HL>{});
            }
#pragma unroll 1
            for (int pp = 0; pp < 5; ++pp) {
                layer(std::integral_constant<int, 3>{}, 2 * pp + 2, std::bool_constant<HL>{});
                layer(std::integral_constant<int, 0>{}, 2 * pp + 3, std::bool_constant<HL>{});
            }
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
