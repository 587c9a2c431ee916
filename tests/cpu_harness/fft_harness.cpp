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
        case 8: Bf<8>::run(*reinterpret_cast<cf(*)[8]>(x)); return 1;
        case 16: Bf<16>::run(*reinterpret_cast<cf(*)[16]>(x)); return 1;
        case 25: Bf<25>::run(*reinterpret_cast<cf(*)[25]>(x)); return 1;
    }
    return 0;
}
