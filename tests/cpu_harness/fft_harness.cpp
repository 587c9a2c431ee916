// Host harness for tests/test_fft_plan_cpu.py: runs the device FFT code of stofnet_amd/csrc/fft_small.h on the CPU
// (one "thread", or several interleaved to exercise the tid/nthreads striding) so the plan, the twiddle tables, the
// butterflies and the fused middle pass can be checked against numpy without a GPU.
#include <vector>
#include "../../stofnet_amd/csrc/fft_small.h"

using namespace stof_fft;

extern "C" int fft_plan(int n, int* radix_out) {
    Plan p;
    if (!make_plan(n, &p)) return 0;
    for (int i = 0; i < p.npass; ++i) radix_out[i] = p.radix[i];
    return p.npass;
}

// z[2n] (re, im interleaved) <- analytic signal transform ifft(H .* fft(z)); returns 0 if n has no plan
extern "C" int fft_analytic(int n, float* z, int nthreads) {
    Plan p;
    if (!make_plan(n, &p)) return 0;
    const int nb = (n + TW_A - 1) / TW_A;
    std::vector<cf> ta(TW_A), tb(nb);
    for (int t = 0; t < TW_A; ++t) ta[t] = mk((float)cos(-2.0 * M_PI * t / n), (float)sin(-2.0 * M_PI * t / n));
    for (int u = 0; u < nb; ++u) {
        const double a = -2.0 * M_PI * (double)u * TW_A / n;
        tb[u] = mk((float)cos(a), (float)sin(a));
    }
    Twiddles tw{ta.data(), tb.data()};
    cf* Z = reinterpret_cast<cf*>(z);
    // a pass is a set of disjoint butterflies, so running the "threads" one after the other between syncs is exact
    const int last = p.npass - 1;
    int m = n;
    for (int s = 0; s < last; ++s) {
        for (int tid = 0; tid < nthreads; ++tid) run_pass<false>(p.radix[s], Z, n, m, tw, tid, nthreads);
        m /= p.radix[s];
    }
    for (int tid = 0; tid < nthreads; ++tid) {
        if (p.radix[last] == 4) middle_pass<4>(Z, n, tid, nthreads);
        else middle_pass<2>(Z, n, tid, nthreads);
    }
    for (int s = last - 1; s >= 0; --s) {
        m *= p.radix[s];
        for (int tid = 0; tid < nthreads; ++tid) run_pass<true>(p.radix[s], Z, n, m, tw, tid, nthreads);
    }
    return 1;
}

// plain forward DFT of one butterfly size (checks Bf<R> against numpy)
extern "C" int fft_butterfly(int R, float* z) {
    cf* x = reinterpret_cast<cf*>(z);
    switch (R) {
        case 2: Bf<2>::run(*reinterpret_cast<cf(*)[2]>(x)); return 1;
        case 3: Bf<3>::run(*reinterpret_cast<cf(*)[3]>(x)); return 1;
        case 4: Bf<4>::run(*reinterpret_cast<cf(*)[4]>(x)); return 1;
        case 5: Bf<5>::run(*reinterpret_cast<cf(*)[5]>(x)); return 1;
        case 6: Bf<6>::run(*reinterpret_cast<cf(*)[6]>(x)); return 1;
        case 8: Bf<8>::run(*reinterpret_cast<cf(*)[8]>(x)); return 1;
        case 10: Bf<10>::run(*reinterpret_cast<cf(*)[10]>(x)); return 1;
        case 15: Bf<15>::run(*reinterpret_cast<cf(*)[15]>(x)); return 1;
        case 16: Bf<16>::run(*reinterpret_cast<cf(*)[16]>(x)); return 1;
        case 20: Bf<20>::run(*reinterpret_cast<cf(*)[20]>(x)); return 1;
        case 25: Bf<25>::run(*reinterpret_cast<cf(*)[25]>(x)); return 1;
    }
    return 0;
}

// ---- compile-time plans: the passes of analytic_ct<N, T> with the T "threads" run one after the other per pass ------
template <int N, int T, int S, int M>
static void host_level(cf* Z, const cf* W) {
    constexpr CtPlan P = ct_plan_for(N);
    if constexpr (S < P.npass) {
        constexpr int R = P.radix[S];
        for (int tid = 0; tid < T; ++tid) ct_pass<N, M, R, false, T>(Z, W, tid);
        host_level<N, T, S + 1, M / R>(Z, W);
        for (int tid = 0; tid < T; ++tid) ct_pass<N, M, R, true, T>(Z, W, tid);
    } else {
        for (int tid = 0; tid < T; ++tid) ct_middle16<N, T>(Z, tid, ct_filter_default<N>());
    }
}

template <int N>
static int run_ct(float* z, int nthreads) {
    constexpr CtPlan P = ct_plan_for(N);
    static constexpr TwTable<P.table> table = make_tw_table<N, P.table>();
    const cf* W = reinterpret_cast<const cf*>(table.w);
    std::vector<cf> slot(ct_slot_entries(N));                       // padded layout: 2 values after every 16
    cf* in = reinterpret_cast<cf*>(z);
    for (int i = 0; i < N; ++i) slot[ct_padded(i)] = in[i];
    cf* Z = slot.data();
    if (nthreads == 64) host_level<N, 64, 0, N>(Z, W);
    else if (nthreads == 128) host_level<N, 128, 0, N>(Z, W);
    else if (nthreads == 256) host_level<N, 256, 0, N>(Z, W);
    else return 0;
    for (int i = 0; i < N; ++i) in[i] = slot[ct_padded(i)];
    return 1;
}

extern "C" int fft_analytic_ct(int n, float* z, int nthreads) {
    switch (n) {
        case 96: return run_ct<96>(z, nthreads);
        case 1536: return run_ct<1536>(z, nthreads);
        case 2000: return run_ct<2000>(z, nthreads);
        case 2048: return run_ct<2048>(z, nthreads);
        case 4000: return run_ct<4000>(z, nthreads);
        case 4096: return run_ct<4096>(z, nthreads);
        case 8000: return run_ct<8000>(z, nthreads);
    }
    return 0;
}

extern "C" int fft_ct_plan(int n, int* radix_out, int* table_out) {
    const CtPlan p = ct_plan_for(n);
    if (p.npass < 0) return -1;
    for (int i = 0; i < p.npass; ++i) radix_out[i] = p.radix[i];
    *table_out = p.table;
    return p.npass;
}

// w_n^k of the compile-time table (checked against numpy)
extern "C" int fft_ct_table(int n, float* out) {
    if (n == 2000) { static constexpr TwTable<400> t = make_tw_table<2000, 400>(); for (int k = 0; k < 400; ++k) { out[2 * k] = t.w[k].x; out[2 * k + 1] = t.w[k].y; } return 400; }
    if (n == 1536) { static constexpr TwTable<256> t = make_tw_table<1536, 256>(); for (int k = 0; k < 256; ++k) { out[2 * k] = t.w[k].x; out[2 * k + 1] = t.w[k].y; } return 256; }
    return 0;
}

// the unpadded layout with the two-level twiddle table (rows whose padded image does not fit LDS)
template <int N, int T, int S, int M>
static void host_level2(cf* Z, const cf* W) {
    using OPT = CtOpt<false, true>;
    constexpr CtPlan P = ct_plan_for(N);
    if constexpr (S < P.npass) {
        constexpr int R = P.radix[S];
        for (int tid = 0; tid < T; ++tid) ct_pass<N, M, R, false, T, OPT>(Z, W, tid);
        host_level2<N, T, S + 1, M / R>(Z, W);
        for (int tid = 0; tid < T; ++tid) ct_pass<N, M, R, true, T, OPT>(Z, W, tid);
    } else {
        for (int tid = 0; tid < T; ++tid) ct_middle16<N, T, OPT>(Z, tid, ct_filter_default<N>());
    }
}
template <int N, int T>
static int run_ct2(float* z) {
    static constexpr auto table = make_tw_table2<N>();
    host_level2<N, T, 0, N>(reinterpret_cast<cf*>(z), reinterpret_cast<const cf*>(table.w));
    return 1;
}
extern "C" int fft_analytic_ct_plain(int n, float* z) {
    switch (n) {
        case 2000: return run_ct2<2000, 64>(z);
        case 8000: return run_ct2<8000, 256>(z);
        case 20000: return run_ct2<20000, 512>(z);
    }
    return 0;
}
