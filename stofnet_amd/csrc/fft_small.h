// Mixed-radix in-place FFT building blocks for the Hilbert / GradPeak kernels (gfx950), written so that the same
// code also compiles for the host (tests/test_fft_plan_cpu.py drives it through a tiny g++-built harness).
//
// Transform of length n = R_0 * R_1 * ... * R_{k-1} on an array Z of n complex values that lives in LDS:
//   forward  : decimation in frequency, pass s works on blocks of m_s = n / (R_0..R_{s-1}) values with radix R_s:
//              butterfly (blk, j) takes Z[blk*m + j + q*sub], sub = m / R, replaces it by
//              y_q = w_m^{j q} * sum_k x_k w_R^{q k}; after the last pass position p holds the frequency whose
//              mixed-radix digits are those of p in reverse order (no permutation pass exists).
//   middle   : the last forward pass, the Hilbert filter (utils/hilbert.py:13-17) and the first inverse pass act on the
//              same R_{k-1} values, so they run back to back in registers.
//   inverse  : decimation in time with the passes in reverse order, x = conj-DFT(conj(w) .* y), consuming the
//              digit-reversed spectrum and producing natural order.  The 1/n lives in the filter.
// The inverse butterflies are the forward ones with the outputs taken in reverse order (IDFT_R(x)[q] = DFT_R(x)[R-q]),
// the inverse twiddles a multiplication by the conjugate (cmulc).
//
// Arithmetic: a complex value is one 64-bit register pair and every primitive below is one or two PACKED fp32
// instructions on gfx950 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel / neg modifiers doing the re/im swaps and
// sign flips of a complex product for free) -- half the VALU issue slots of scalar code, which is what bounds these
// kernels.  The host build (tests/cpu_harness) runs the same formulas in scalar code, rounding for rounding.
//
// Two front ends share the butterflies: run-time plans (any smooth n, `analytic_in_place`) and compile-time plans
// (`analytic_ct<N, T>`: radices, strides, trip counts and LDS offsets are constants, twiddles come from a full table
// w_N^t built at compile time in double precision and rounded once).
//
// Twiddles w_n^t come from a two-level table built per work-group in double precision and rounded once:
//   w_n^t = TB[t >> TW_SHIFT] * TA[t & (TW_A - 1)]            (one fp32 product)
// and the R-1 powers a butterfly needs are the binary powers w^{t}, w^{2t}, w^{4t}, ... looked up exactly and the rest
// formed with at most three products (phase error <= 4 roundings).
#pragma once
#include <math.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define STOF_HD __host__ __device__ __forceinline__
#else
#define STOF_HD inline
#endif

namespace stof_fft {

#if defined(__HIPCC__)
// one aligned VGPR pair: (re, im).  may_alias: the kernels move the same LDS bytes as 16-byte pieces (pair_io.h) and as
// floats (GradPeak streams envelopes out of the transform's buffer); without it type-based alias analysis lets the
// compiler move a complex read across such a store (seen: the first pass of n = 1536 read rows before they were staged).
typedef float cf __attribute__((ext_vector_type(2), may_alias));
#else
struct alignas(8) cf {
    float x, y;
};
#endif
STOF_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }

// Ordering point for LDS that ONE wavefront shares among its lanes.  The hardware executes a wave's LDS instructions
// in order, so no s_barrier / s_waitcnt is needed, but the compiler reasons per thread: it may move a lane's read of
// ANOTHER lane's slot above its own earlier store to a different address (seen: the first pass of n = 1536 was
// scheduled before the last row pieces were staged).  A wavefront-scope release / acquire pair around a wave barrier
// pins the order and emits no instruction.
STOF_HD void wave_lds_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
STOF_HD cf cadd(cf a, cf b) { return a + b; }
STOF_HD cf csub(cf a, cf b) { return a - b; }
STOF_HD cf cscale(cf a, float s) { return a * s; }
STOF_HD cf fma_real(float c, cf b, cf a) { return __builtin_elementwise_fma((cf)(c), b, a); }      // a + c b, c real
// a * w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x): t = (-a.y w.y, a.y w.x), r = a.xx * w + t
STOF_HD cf cmul(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w) = (a.x w.x + a.y w.y, -a.x w.y + a.y w.x): t = (a.y w.y, a.y w.x), r = a.xx * (w.x, -w.y) + t
STOF_HD cf cmulc(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// the same product with a wave-uniform w (a literal root of unity): w rides a scalar register pair
STOF_HD cf cmul_uniform(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
// a + i v = (a.x - v.y, a.y + v.x) and a - i v = (a.x + v.y, a.y - v.x)
STOF_HD cf add_i(cf a, cf v) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(v));
    return r;
}
STOF_HD cf sub_i(cf a, cf v) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(v));
    return r;
}
#else
STOF_HD cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
STOF_HD cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
STOF_HD cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
STOF_HD cf fma_real(float c, cf b, cf a) { return mk(fmaf(c, b.x, a.x), fmaf(c, b.y, a.y)); }
STOF_HD cf cmul(cf a, cf w) { return mk(fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.x, w.y, a.y * w.x)); }
STOF_HD cf cmulc(cf a, cf w) { return mk(fmaf(a.x, w.x, a.y * w.y), fmaf(a.x, -w.y, a.y * w.x)); }
STOF_HD cf cmul_uniform(cf a, cf w) { return cmul(a, w); }
STOF_HD cf add_i(cf a, cf v) { return mk(a.x - v.y, a.y + v.x); }
STOF_HD cf sub_i(cf a, cf v) { return mk(a.x + v.y, a.y - v.x); }
#endif

// ---- compile-time roots of unity ------------------------------------------------------------------------------
// cos / sin of 2 pi k / N in double by octant reduction + Taylor series (constexpr, so the constants are literals).
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double ct_sin_small(double x) {       // |x| <= pi/4
    double term = x, sum = x;
    for (int i = 1; i < 14; ++i) { term *= -x * x / ((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
constexpr double ct_cos_small(double x) {
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < 14; ++i) { term *= -x * x / ((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
struct cd { double c, s; };
constexpr cd ct_root(int k, int N) {            // exp(-2 pi i k / N) as (cos, -sin)
    k %= N; if (k < 0) k += N;
    // reduce the angle a = 2 pi k / N to an octant using exact integer arithmetic on 8k / N
    const int oct = (8 * k) / N;                 // 0..7
    const double a = 2.0 * kPi * (double)k / (double)N;
    double c = 0, s = 0;
    switch (oct) {
        case 0: c = ct_cos_small(a); s = ct_sin_small(a); break;
        case 1: c = ct_sin_small(kPi / 2 - a); s = ct_cos_small(kPi / 2 - a); break;
        case 2: c = -ct_sin_small(a - kPi / 2); s = ct_cos_small(a - kPi / 2); break;
        case 3: c = -ct_cos_small(kPi - a); s = ct_sin_small(kPi - a); break;
        case 4: c = -ct_cos_small(a - kPi); s = -ct_sin_small(a - kPi); break;
        case 5: c = -ct_sin_small(3 * kPi / 2 - a); s = -ct_cos_small(3 * kPi / 2 - a); break;
        case 6: c = ct_sin_small(a - 3 * kPi / 2); s = -ct_cos_small(a - 3 * kPi / 2); break;
        default: c = ct_cos_small(2 * kPi - a); s = -ct_sin_small(2 * kPi - a); break;
    }
    return cd{c, -s};
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <class F, int... I>
STOF_HD void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
STOF_HD void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// v *= exp(-2 pi i E / R), E and R compile-time: the trivial roots cost no multiply, the others are literal constants
template <int E0, int R>
STOF_HD void mul_root(cf& v) {
    constexpr int E = ((E0 % R) + R) % R;
    if constexpr (E == 0) {
    } else if constexpr (4 * E == R) {
        v = sub_i(mk(0.f, 0.f), v);                               // -i v
    } else if constexpr (2 * E == R) {
        v = csub(mk(0.f, 0.f), v);                                // -v
    } else if constexpr (4 * E == 3 * R) {
        v = add_i(mk(0.f, 0.f), v);                               // +i v
    } else {
        constexpr cd w = ct_root(E, R);
        v = cmul_uniform(v, mk((float)w.c, (float)w.s));
    }
}

// ---- forward butterflies (y_q = sum_k x_k exp(-2 pi i q k / R)), in place, natural order in and out --------------
template <int R> struct Bf;

template <> struct Bf<2> {
    static STOF_HD void run(cf (&x)[2]) {
        const cf a = x[0], b = x[1];
        x[0] = cadd(a, b); x[1] = csub(a, b);
    }
};
template <> struct Bf<4> {
    static STOF_HD void run(cf (&x)[4]) {
        const cf s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]);
        const cf s13 = cadd(x[1], x[3]), d13 = csub(x[1], x[3]);
        x[0] = cadd(s02, s13); x[1] = sub_i(d02, d13); x[2] = csub(s02, s13); x[3] = add_i(d02, d13);   // -+ i d13
    }
};
template <> struct Bf<3> {
    static STOF_HD void run(cf (&x)[3]) {
        const cf s12 = cadd(x[1], x[2]), d12 = csub(x[1], x[2]);
        const cf t = fma_real(-0.5f, s12, x[0]);
        constexpr float sn = -0.8660254037844386f;               // -sin(2 pi / 3)
        const cf v = cscale(d12, sn);
        x[0] = cadd(x[0], s12); x[1] = add_i(t, v); x[2] = sub_i(t, v);       // t +- i sn d12
    }
};
template <> struct Bf<5> {
    static STOF_HD void run(cf (&x)[5]) {
        constexpr float c1 = 0.30901699437494745f, c2 = -0.8090169943749475f;
        constexpr float s1 = -0.9510565162951535f, s2 = -0.5877852522924731f;
        const cf a14 = cadd(x[1], x[4]), b14 = csub(x[1], x[4]);
        const cf a23 = cadd(x[2], x[3]), b23 = csub(x[2], x[3]);
        const cf t1 = fma_real(c2, a23, fma_real(c1, a14, x[0]));
        const cf t2 = fma_real(c1, a23, fma_real(c2, a14, x[0]));
        const cf v1 = fma_real(s2, b23, cscale(b14, s1));         // x_1, x_4 = t1 +- i v1
        const cf v2 = fma_real(-s1, b23, cscale(b14, s2));        // x_2, x_3 = t2 +- i v2
        x[0] = cadd(x[0], cadd(a14, a23));
        x[1] = add_i(t1, v1); x[4] = sub_i(t1, v1); x[2] = add_i(t2, v2); x[3] = sub_i(t2, v2);
    }
};

// R = A * B by Cooley-Tukey in registers: B-point DFTs over n2 (stride A), twiddle w_R^{n1 k2}, A-point DFTs over n1,
// then the (compile-time) digit permutation back to natural order.
template <int A, int B> struct BfComposite {
    static constexpr int R = A * B;
    static STOF_HD void run(cf (&x)[R]) {
        cf y[R];
        static_for<A>([&](auto n1c) {
            constexpr int n1 = decltype(n1c)::value;
            cf t[B];
            static_for<B>([&](auto n2c) { constexpr int n2 = decltype(n2c)::value; t[n2] = x[n1 + A * n2]; });
            Bf<B>::run(t);
            static_for<B>([&](auto k2c) {
                constexpr int k2 = decltype(k2c)::value;
                cf v = t[k2];
                mul_root<n1 * k2, R>(v);
                y[n1 + A * k2] = v;
            });
        });
        static_for<B>([&](auto k2c) {
            constexpr int k2 = decltype(k2c)::value;
            cf t[A];
            static_for<A>([&](auto n1c) { constexpr int n1 = decltype(n1c)::value; t[n1] = y[n1 + A * k2]; });
            Bf<A>::run(t);
            static_for<A>([&](auto k1c) { constexpr int k1 = decltype(k1c)::value; x[B * k1 + k2] = t[k1]; });
        });
    }
};
template <> struct Bf<6> { static STOF_HD void run(cf (&x)[6]) { BfComposite<2, 3>::run(x); } };
template <> struct Bf<10> { static STOF_HD void run(cf (&x)[10]) { BfComposite<2, 5>::run(x); } };
template <> struct Bf<8> { static STOF_HD void run(cf (&x)[8]) { BfComposite<2, 4>::run(x); } };
template <> struct Bf<15> { static STOF_HD void run(cf (&x)[15]) { BfComposite<3, 5>::run(x); } };
template <> struct Bf<16> { static STOF_HD void run(cf (&x)[16]) { BfComposite<4, 4>::run(x); } };
template <> struct Bf<20> { static STOF_HD void run(cf (&x)[20]) { BfComposite<4, 5>::run(x); } };
template <> struct Bf<25> { static STOF_HD void run(cf (&x)[25]) { BfComposite<5, 5>::run(x); } };

// ---- plan ---------------------------------------------------------------------------------------------------------
constexpr int MAX_PASSES = 12;
constexpr int TW_SHIFT = 6;
constexpr int TW_A = 1 << TW_SHIFT;

struct Plan {
    int n;
    int npass;                 // including the middle pass (the last entry)
    int radix[MAX_PASSES];     // radix[npass-1] is 2 or 4
};

// n even with only the factors 2, 3, 5 (and small enough for the caller's LDS budget): radices taken greedily from
// {16, 8, 5, 4, 3, 2}, the middle radix (4 if 4 | n else 2) last.  Returns false if n has another prime factor.
// (A radix-25 butterfly saves a pass at n = 2000 but needs 180 VGPRs; without it the kernels stay under 128 and run
// four waves per SIMD, which is worth more: these kernels are LDS-latency bound, not arithmetic bound.)
constexpr Plan plan_for(int n) {                 // npass = 0: no plan
    Plan p{};
    p.n = n;
    if (n < 2 || (n & 1)) return p;
    const int mid = (n % 4 == 0) ? 4 : 2;
    int m = n / mid;
    const int cand[6] = {16, 8, 5, 4, 3, 2};
    while (m > 1) {
        int r = 0;
        for (int i = 0; i < 6 && !r; ++i) if (m % cand[i] == 0) r = cand[i];
        if (!r || p.npass >= MAX_PASSES - 1) { p.npass = 0; return p; }
        p.radix[p.npass++] = r;
        m /= r;
    }
    p.radix[p.npass++] = mid;
    return p;
}
inline bool make_plan(int n, Plan* p) {
    *p = plan_for(n);
    return p->npass > 0;
}

// ---- twiddle tables (LDS): TA[t] = w_n^t, t < TW_A;  TB[u] = w_n^{u * TW_A}, u < ceil(n / TW_A) ----------------------
struct Twiddles {
    const cf* ta;
    const cf* tb;
    STOF_HD cf at(int t) const { return cmul(tb[t >> TW_SHIFT], ta[t & (TW_A - 1)]); }   // tb[0] = 1: exact for t < TW_A
};
inline int twiddle_entries(int n) { return TW_A + (n + TW_A - 1) / TW_A; }

// x[q] *= w_n^{q t1}, q = 1 .. R-1 (t1 * (R-1) < n).  The binary powers w^{t1}, w^{2 t1}, w^{4 t1}, ... are looked up
// exactly; w^{q t1} = w^{(q minus its lowest set bit) t1} * w^{(lowest set bit) t1} costs one product and is used at
// once, so only the powers along q's bit prefix are alive (registers: ~2 log2 R complex values instead of R).
template <int R, bool CONJ>
STOF_HD void apply_twiddles(const Twiddles& tw, int t1, cf (&x)[R]) {
    constexpr int NB = R <= 2 ? 1 : R <= 4 ? 2 : R <= 8 ? 3 : R <= 16 ? 4 : 5;
    cf bin[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) bin[k] = ((1 << k) < R) ? tw.at(t1 << k) : mk(1.f, 0.f);
    cf w[R];
    static_for<R>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q >= 1) {
            constexpr int low = q & -q, rest = q - low;
            constexpr int k = low == 1 ? 0 : low == 2 ? 1 : low == 4 ? 2 : low == 8 ? 3 : 4;
            if constexpr (rest == 0) w[q] = bin[k];
            else w[q] = cmul(w[rest], bin[k]);
            x[q] = CONJ ? cmulc(x[q], w[q]) : cmul(x[q], w[q]);
        }
    });
}

// b / sub for b < 2^22 without an integer division (float quotient + one correction step)
STOF_HD void divmod_small(int b, int sub, float inv_sub, int& q, int& r) {
    q = (int)((float)b * inv_sub);
    r = b - q * sub;
    if (r < 0) { r += sub; q -= 1; } else if (r >= sub) { r -= sub; q += 1; }
}

// One forward (DIF) or inverse (DIT) pass over Z[0, n): threads tid, tid + nthreads, ... each take one butterfly.
template <int R, bool INV>
STOF_HD void pass(cf* Z, int n, int m, const Twiddles& tw, int tid, int nthreads) {
    const int sub = m / R, tstep = n / m, nb = n / R;
    const float inv_sub = 1.0f / (float)sub;
    for (int b = tid; b < nb; b += nthreads) {
        int blk, j;
        divmod_small(b, sub, inv_sub, blk, j);
        cf* base = Z + blk * m + j;
        cf x[R];
#pragma unroll
        for (int k = 0; k < R; ++k) x[k] = base[k * sub];
        if (INV) {                                               // j = 0 gives exact ones: no special case
            apply_twiddles<R, true>(tw, j * tstep, x);
            Bf<R>::run(x);
#pragma unroll
            for (int k = 0; k < R; ++k) base[k * sub] = x[(R - k) % R];       // inverse DFT = forward, outputs reversed
        } else {
            Bf<R>::run(x);
            apply_twiddles<R, false>(tw, j * tstep, x);
#pragma unroll
            for (int k = 0; k < R; ++k) base[k * sub] = x[k];
        }
    }
}

// Middle pass: forward radix-RM butterfly on RM adjacent values, the Hilbert filter H[k] / n, inverse butterfly.
// Position blk*RM + q holds frequency k = k_low(blk) + q * n/RM with k_low(blk) = 0 iff blk = 0, so for even n
//   RM = 4: q = 0 -> 2/n (1/n at k = 0), q = 1 -> 2/n, q = 2 -> 1/n at the Nyquist bin (blk = 0) else 0, q = 3 -> 0
//   RM = 2: q = 0 -> 2/n (1/n at k = 0), q = 1 -> 1/n at the Nyquist bin (blk = 0) else 0.
template <int RM>
STOF_HD void middle_pass(cf* Z, int n, int tid, int nthreads) {
    const int nb = n / RM;
    const float one = 1.0f / (float)n, two = 2.0f / (float)n;
    for (int b = tid; b < nb; b += nthreads) {
        cf* base = Z + b * RM;
        cf x[RM];
#pragma unroll
        for (int k = 0; k < RM; ++k) x[k] = base[k];
        Bf<RM>::run(x);
        const bool first = (b == 0);
        if (RM == 4) {
            x[0] = cscale(x[0], first ? one : two);
            x[1] = cscale(x[1], two);
            x[2] = cscale(x[2], first ? one : 0.f);
            x[3] = mk(0.f, 0.f);
        } else {
            x[0] = cscale(x[0], first ? one : two);
            x[1] = cscale(x[1], first ? one : 0.f);
        }
        Bf<RM>::run(x);
#pragma unroll
        for (int k = 0; k < RM; ++k) base[k] = x[(RM - k) % RM];
    }
}

template <bool INV>
STOF_HD void run_pass(int R, cf* Z, int n, int m, const Twiddles& tw, int tid, int nthreads) {
    switch (R) {
        case 2: pass<2, INV>(Z, n, m, tw, tid, nthreads); break;
        case 3: pass<3, INV>(Z, n, m, tw, tid, nthreads); break;
        case 4: pass<4, INV>(Z, n, m, tw, tid, nthreads); break;
        case 5: pass<5, INV>(Z, n, m, tw, tid, nthreads); break;
        case 8: pass<8, INV>(Z, n, m, tw, tid, nthreads); break;
        default: pass<16, INV>(Z, n, m, tw, tid, nthreads); break;
    }
}

// analytic signal of the n complex values in Z, in place: Z <- ifft(H .* fft(Z)).  `sync()` separates the passes
// (a work-group barrier, or wave_lds_sync() when one wave owns Z).
template <class Sync>
STOF_HD void analytic_in_place(cf* Z, const Plan& plan, const Twiddles& tw, int tid, int nthreads, Sync sync) {
    const int n = plan.n, last = plan.npass - 1;
    int m = n;
    for (int s = 0; s < last; ++s) {
        run_pass<false>(plan.radix[s], Z, n, m, tw, tid, nthreads);
        m /= plan.radix[s];
        sync();
    }
    if (plan.radix[last] == 4) middle_pass<4>(Z, n, tid, nthreads);
    else middle_pass<2>(Z, n, tid, nthreads);
    sync();
    for (int s = last - 1; s >= 0; --s) {
        m *= plan.radix[s];
        run_pass<true>(plan.radix[s], Z, n, m, tw, tid, nthreads);
        sync();
    }
}


// ---- compile-time plans --------------------------------------------------------------------------------------------
// For N = 16 * R_0 * ... * R_{k-1} (R_s in {16, 10, 8, 6, 5, 4, 3, 2}).  Everything a run-time plan computes per
// butterfly (block / offset split, strides, twiddle indices, trip counts) is a constant here, and the structure is
// shaped by what the LDS of gfx950 costs (MI355X guide, LDS section: ds_write_b64 ~6 cycles per wave instruction,
// bank conflicts serialise a 32-lane group):
//   * the last 16 values of the decomposition are one in-register radix-16 MIDDLE pass (forward DFT_16, Hilbert
//     filter, inverse DFT_16 pruned of its zero inputs): an LDS round trip per pass less on each side;
//   * Z is stored with two complex values of padding after every 16 (`pad`), so that a thread's 128 contiguous bytes of
//     the middle pass, and the 16-byte pieces of the row staging, fall on distinct banks; all other passes have
//     strides that are multiples of 16 values, so their offsets stay compile-time constants in the padded layout;
//   * one twiddle read per butterfly, w_M^j from a table of the first N / min R_s powers of w_N (compile-time, double
//     precision, rounded once), the other R-2 factors by products (w^2q = (w^q)^2, w^(2q+1) = w^2q w: <= 2 log2 R
//     roundings) -- a strided read of the table conflicts on its banks, a packed product costs two VALU slots;
//   * each pass is software pipelined over groups of U butterflies per thread: group g+1 is read from LDS before group
//     g is computed and stored (the compiler keeps reads of Z behind earlier writes to Z, so the order is set here).
struct f2c { float x, y; };
template <int K> struct TwTable { f2c w[K]; };
template <int N, int K>
constexpr TwTable<K> make_tw_table() {            // w_N^t, t < K: double-precision roots, rounded once
    TwTable<K> t{};
    for (int k = 0; k < K; ++k) {
        const cd r = ct_root(k, N);
        t.w[k].x = (float)r.c;
        t.w[k].y = (float)r.s;
    }
    return t;
}

struct CtPlan {
    int npass;                  // passes before the radix-16 middle pass
    int radix[MAX_PASSES];
    int table;                  // twiddle table entries: N / min radix
};
constexpr CtPlan ct_plan_for(int n) {             // npass < 0: no plan
    CtPlan p{};
    if (n < 16 || n % 16 != 0) { p.npass = -1; return p; }
    int m = n / 16, rmin = 1 << 30;
    const int cand[8] = {16, 10, 8, 6, 5, 4, 3, 2};
    while (m > 1) {
        int r = 0;
        for (int i = 0; i < 8 && !r; ++i) if (m % cand[i] == 0) r = cand[i];
        if (!r || p.npass >= MAX_PASSES - 1) { p.npass = -1; return p; }
        p.radix[p.npass++] = r;
        if (r < rmin) rmin = r;
        m /= r;
    }
    p.table = p.npass ? n / rmin : 1;
    return p;
}
constexpr int ct_padded(int i) { return i + 2 * (i >> 4); }            // element index in the padded layout
constexpr int ct_slot_entries(int n) { return ct_padded(n); }          // complex values per row slot (n % 16 == 0)
STOF_HD unsigned pad(unsigned i) { return i + 2u * (i >> 4); }

// Options of a compile-time plan.  PAD: the padded layout above (default).  Rows whose padded image would not fit LDS
// (20,000 values = 160,000 bytes) run unpadded: only the middle pass pays bank conflicts then.  TW2: the twiddle table
// is the two-level pair of fft_small's run-time plans, w_N^t = TB[t >> 6] * TA[t & 63] (TA at W[0..63], TB behind it),
// when the full table of N / min-radix entries does not fit next to the image either.
template <bool PAD_ = true, bool TW2_ = false>
struct CtOpt {
    static constexpr bool PAD = PAD_, TW2 = TW2_;
};
template <int N> constexpr int ct_tw2_entries() { return TW_A + (ct_plan_for(N).table + TW_A - 1) / TW_A; }
template <int N>
constexpr TwTable<ct_tw2_entries<N>()> make_tw_table2() {      // TA[t] = w_N^t (t < 64), then TB[u] = w_N^{64 u}
    TwTable<ct_tw2_entries<N>()> t{};
    for (int k = 0; k < ct_tw2_entries<N>(); ++k) {
        const cd r = ct_root(k < TW_A ? k : (k - TW_A) * TW_A, N);
        t.w[k].x = (float)r.c;
        t.w[k].y = (float)r.s;
    }
    return t;
}

template <int R, int UU>
struct CtGroup {
    cf x[UU][R];
    cf w1[UU];
    unsigned off[UU];        // padded element offset of the butterfly's first value
    bool on[UU];
};

// pipelined execution of NG groups: load(G, g) / finish(G); groups 0 .. NG-2 alternate between two register sets of U
// butterflies, the last group has ULAST <= U
template <int R, int U, int ITERS, class Load, class Finish>
STOF_HD void ct_pipeline(Load load, Finish finish) {
    constexpr int NG = (ITERS + U - 1) / U, ULAST = ITERS - (NG - 1) * U;
    CtGroup<R, U> A, B;
    CtGroup<R, ULAST> L;
    if constexpr (NG == 1) {
        load(L, std::integral_constant<int, 0>{});
        finish(L);
    } else {
        load(A, std::integral_constant<int, 0>{});
        static_for<NG - 1>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < NG - 1) {
                if constexpr (g % 2 == 0) load(B, std::integral_constant<int, g + 1>{});
                else load(A, std::integral_constant<int, g + 1>{});
            } else {
                load(L, std::integral_constant<int, NG - 1>{});
            }
            if constexpr (g % 2 == 0) finish(A);
            else finish(B);
        });
        finish(L);
    }
}

// One pass of radix R over blocks of M values (M / R a multiple of 16) by T threads; W[t] = w_N^t.
template <int N, int M, int R, bool INV, int T, class OPT = CtOpt<>>
STOF_HD void ct_pass(cf* __restrict__ Z, const cf* __restrict__ W, int tid) {
    constexpr int SUB = M / R, TSTEP = N / M, NB = N / R, ITERS = (NB + T - 1) / T;
    constexpr int SUBP = OPT::PAD ? ct_padded(SUB) : SUB, MP = OPT::PAD ? ct_padded(M) : M;
    constexpr int U = R >= 8 ? 1 : 2;
    constexpr bool RAGGED = (NB % T) != 0;
    static_assert(SUB % 16 == 0 && N % M == 0 && M % R == 0, "bad pass geometry");
    auto load = [&](auto& G, auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int UU = sizeof(G.on) / sizeof(bool);
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const unsigned b = (unsigned)(tid + (g * U + u) * T);
            G.on[u] = !(RAGGED && g * U + u == ITERS - 1) || (int)b < NB;      // only the last iteration is ragged
            const unsigned bb = G.on[u] ? b : 0u;
            const unsigned blk = (M == N) ? 0u : bb / (unsigned)SUB;
            const unsigned j = (M == N) ? bb : bb - blk * (unsigned)SUB;
            G.off[u] = blk * (unsigned)MP + (OPT::PAD ? pad(j) : j);
#pragma unroll
            for (int k = 0; k < R; ++k) G.x[u][k] = Z[G.off[u] + k * SUBP];
            const unsigned t = j * (unsigned)TSTEP;                            // w_M^j = w_N^{j TSTEP}
            if constexpr (OPT::TW2) G.w1[u] = cmul(W[TW_A + (t >> TW_SHIFT)], W[t & (TW_A - 1)]);
            else G.w1[u] = W[t];
        }
    };
    auto finish = [&](auto& G) {
        constexpr int UU = sizeof(G.on) / sizeof(bool);
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            if (!INV) Bf<R>::run(G.x[u]);
            cf w[R];
            w[1] = G.w1[u];
            static_for<R>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q >= 2) w[q] = (q % 2 == 0) ? cmul(w[q / 2], w[q / 2]) : cmul(w[q - 1], w[1]);
                if constexpr (q >= 1) G.x[u][q] = INV ? cmulc(G.x[u][q], w[q]) : cmul(G.x[u][q], w[q]);
            });
            if (INV) Bf<R>::run(G.x[u]);
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            if (G.on[u]) {
#pragma unroll
                for (int k = 0; k < R; ++k) Z[G.off[u] + k * SUBP] = G.x[u][INV ? (R - k) % R : k];
            }
        }
    };
    ct_pipeline<R, U, ITERS>(load, finish);
}

// Middle pass: block b = 16 adjacent values (128 contiguous bytes, 144-byte pitch) holds, after the forward passes, the
// samples whose DFT_16 gives the frequencies k = k_low(b) + q N/16, q = 0..15, with k_low(b) = 0 iff b = 0.  Filter
// (utils/hilbert.py:13-17 with the inverse transform's 1/N): q < 8 -> 2/N (1/N at k = 0), q = 8 -> 1/N at the Nyquist
// bin (b = 0) else 0, q > 8 -> 0; then the inverse DFT_16.
// CtFilter: the scale of bins below Nyquist (`two`), of the DC and Nyquist bins (`one`), and whether this transform holds
// those two bins at all (`edge`).  A stand-alone transform of length N uses {1/N, 2/N, true}; the inner blocks of a
// four-step transform of length R0 * N hold the frequencies q + R0 k, so block q != 0 has neither DC nor Nyquist and all
// blocks scale by the full length.
struct CtFilter { float one, two; bool edge; };
template <int N> STOF_HD CtFilter ct_filter_default() {
    CtFilter f; f.one = (float)(1.0 / (double)N); f.two = (float)(2.0 / (double)N); f.edge = true; return f;
}

template <int N, int T, class OPT = CtOpt<>>
STOF_HD void ct_middle16(cf* __restrict__ Z, int tid, const CtFilter filt) {
    constexpr int NB = N / 16, ITERS = (NB + T - 1) / T;
    constexpr bool RAGGED = (NB % T) != 0;
    const float one = filt.one, two = filt.two;
    auto load = [&](auto& G, auto gc) {
        constexpr int g = decltype(gc)::value;
        const unsigned b = (unsigned)(tid + g * T);
        G.on[0] = !(RAGGED && g == ITERS - 1) || (int)b < NB;
        G.off[0] = (G.on[0] ? b : 0u) * (OPT::PAD ? 18u : 16u);
#pragma unroll
        for (int k = 0; k < 16; ++k) G.x[0][k] = Z[G.off[0] + k];
    };
    auto finish = [&](auto& G) {
        const bool first = (G.off[0] == 0) && filt.edge;
        Bf<16>::run(G.x[0]);
        G.x[0][0] = cscale(G.x[0][0], first ? one : two);
#pragma unroll
        for (int q = 1; q < 8; ++q) G.x[0][q] = cscale(G.x[0][q], two);
        G.x[0][8] = cscale(G.x[0][8], first ? one : 0.f);
#pragma unroll
        for (int q = 9; q < 16; ++q) G.x[0][q] = mk(0.f, 0.f);
        Bf<16>::run(G.x[0]);
        if (G.on[0]) {
#pragma unroll
            for (int k = 0; k < 16; ++k) Z[G.off[0] + k] = G.x[0][(16 - k) % 16];
        }
    };
    ct_pipeline<16, 1, ITERS>(load, finish);
}

// forward passes on the way down, the middle pass at the bottom, the inverse passes (reverse order) on the way up
template <int N, int T, int S, int M, class OPT, class Sync>
STOF_HD void ct_level(cf* __restrict__ Z, const cf* __restrict__ W, int tid, Sync& sync, const CtFilter& filt) {
    constexpr CtPlan P = ct_plan_for(N);
    static_assert(P.npass >= 0, "no compile-time plan for this length");
    if constexpr (S < P.npass) {
        constexpr int R = P.radix[S];
        ct_pass<N, M, R, false, T, OPT>(Z, W, tid);
        sync();
        ct_level<N, T, S + 1, M / R, OPT>(Z, W, tid, sync, filt);
        ct_pass<N, M, R, true, T, OPT>(Z, W, tid);
        sync();
    } else {
        static_assert(M == 16, "the middle pass takes the last 16 values");
        ct_middle16<N, T, OPT>(Z, tid, filt);
        sync();
    }
}

// analytic signal of the N complex values in the padded slot Z, in place, by T threads (tid < T)
template <int N, int T, class OPT = CtOpt<>, class Sync>
STOF_HD void analytic_ct(cf* __restrict__ Z, const cf* __restrict__ W, int tid, Sync sync, const CtFilter filt) {
    ct_level<N, T, 0, N, OPT>(Z, W, tid, sync, filt);
}
template <int N, int T, class OPT = CtOpt<>, class Sync>
STOF_HD void analytic_ct(cf* __restrict__ Z, const cf* __restrict__ W, int tid, Sync sync) {
    analytic_ct<N, T, OPT>(Z, W, tid, sync, ct_filter_default<N>());
}

}  // namespace stof_fft
