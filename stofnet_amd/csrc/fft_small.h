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
// The inverse butterflies are the forward ones applied to re/im-swapped values (swap(z)*w = swap(z*conj(w))).
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

struct alignas(8) cf {
    float x, y;
};
STOF_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
STOF_HD cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
STOF_HD cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
STOF_HD cf cmul(cf a, cf b) { return mk(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
STOF_HD cf cswap(cf a) { return mk(a.y, a.x); }
STOF_HD cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }

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
        v = mk(v.y, -v.x);                                        // -i
    } else if constexpr (2 * E == R) {
        v = mk(-v.x, -v.y);                                       // -1
    } else if constexpr (4 * E == 3 * R) {
        v = mk(-v.y, v.x);                                        // +i
    } else {
        constexpr cd w = ct_root(E, R);
        v = cmul(v, mk((float)w.c, (float)w.s));
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
        const cf rot = mk(d13.y, -d13.x);                       // -i * d13
        x[0] = cadd(s02, s13); x[1] = cadd(d02, rot); x[2] = csub(s02, s13); x[3] = csub(d02, rot);
    }
};
template <> struct Bf<3> {
    static STOF_HD void run(cf (&x)[3]) {
        const cf s12 = cadd(x[1], x[2]), d12 = csub(x[1], x[2]);
        const cf t = mk(fmaf(-0.5f, s12.x, x[0].x), fmaf(-0.5f, s12.y, x[0].y));
        constexpr float sn = -0.8660254037844386f;               // -sin(2 pi / 3)
        const cf rot = mk(-sn * d12.y, sn * d12.x);              // i * sn * d12
        x[0] = cadd(x[0], s12); x[1] = cadd(t, rot); x[2] = csub(t, rot);
    }
};
template <> struct Bf<5> {
    static STOF_HD void run(cf (&x)[5]) {
        constexpr float c1 = 0.30901699437494745f, c2 = -0.8090169943749475f;
        constexpr float s1 = -0.9510565162951535f, s2 = -0.5877852522924731f;
        const cf a14 = cadd(x[1], x[4]), b14 = csub(x[1], x[4]);
        const cf a23 = cadd(x[2], x[3]), b23 = csub(x[2], x[3]);
        const cf t1 = mk(x[0].x + c1 * a14.x + c2 * a23.x, x[0].y + c1 * a14.y + c2 * a23.y);
        const cf t2 = mk(x[0].x + c2 * a14.x + c1 * a23.x, x[0].y + c2 * a14.y + c1 * a23.y);
        const cf u1 = mk(-(s1 * b14.y + s2 * b23.y), s1 * b14.x + s2 * b23.x);
        const cf u2 = mk(-(s2 * b14.y - s1 * b23.y), s2 * b14.x - s1 * b23.x);
        x[0] = cadd(x[0], cadd(a14, a23));
        x[1] = cadd(t1, u1); x[4] = csub(t1, u1); x[2] = cadd(t2, u2); x[3] = csub(t2, u2);
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
template <> struct Bf<8> { static STOF_HD void run(cf (&x)[8]) { BfComposite<2, 4>::run(x); } };
template <> struct Bf<16> { static STOF_HD void run(cf (&x)[16]) { BfComposite<4, 4>::run(x); } };
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
inline bool make_plan(int n, Plan* p) {
    p->n = n; p->npass = 0;
    if (n < 2 || (n & 1)) return false;
    const int mid = (n % 4 == 0) ? 4 : 2;
    int m = n / mid;
    static const int cand[6] = {16, 8, 5, 4, 3, 2};
    while (m > 1) {
        int r = 0;
        for (int c : cand) if (m % c == 0) { r = c; break; }
        if (!r || p->npass >= MAX_PASSES - 1) return false;
        p->radix[p->npass++] = r;
        m /= r;
    }
    p->radix[p->npass++] = mid;
    return true;
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
template <int R>
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
            x[q] = cmul(x[q], w[q]);
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
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] = cswap(x[k]);
            apply_twiddles<R>(tw, j * tstep, x);
            Bf<R>::run(x);
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] = cswap(x[k]);
        } else {
            Bf<R>::run(x);
            apply_twiddles<R>(tw, j * tstep, x);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) base[k * sub] = x[k];
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
#pragma unroll
        for (int k = 0; k < RM; ++k) x[k] = cswap(x[k]);
        Bf<RM>::run(x);
#pragma unroll
        for (int k = 0; k < RM; ++k) base[k] = cswap(x[k]);
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
// (a work-group barrier, or nothing when one wave owns Z).
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

}  // namespace stof_fft
