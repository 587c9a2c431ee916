// Standalone check of stof_hilbert (no torch): envelopes of R rows of length n against a double-precision DFT-based
// Hilbert transform on the host.   hipcc -O3 -std=c++17 -fconstexpr-steps=100000000 --offload-arch=gfx950 tools/micro/hilbert_selftest.hip -o tools/micro/hilbert_selftest.bin
#include "../../stofnet_amd/csrc/hilbert.hip"
#include <cstdio>
#include <cmath>
#include <vector>
#include <complex>

static std::vector<double> ref_envelope(const float* x, int n) {
    std::vector<std::complex<double>> X(n), v(n);
    for (int k = 0; k < n; ++k) {
        std::complex<double> s = 0;
        for (int i = 0; i < n; ++i) s += (double)x[i] * std::polar(1.0, -2.0 * M_PI * ((long long)i * k % n) / n);
        double h = (k == 0 || (n % 2 == 0 && k == n / 2)) ? 1.0 : (k < (n + 1) / 2 ? 2.0 : 0.0);
        X[k] = s * h;
    }
    std::vector<double> e(n);
    for (int i = 0; i < n; ++i) {
        std::complex<double> s = 0;
        for (int k = 0; k < n; ++k) s += X[k] * std::polar(1.0, 2.0 * M_PI * ((long long)i * k % n) / n);
        e[i] = std::abs(s / (double)n);
    }
    return e;
}

int main(int argc, char** argv) {
    int bad = 0;
    for (int n : {1536, 2000, 2048}) {
        for (int rows : {1, 2, 5}) {
            std::vector<float> h((size_t)rows * n);
            for (size_t i = 0; i < h.size(); ++i) h[i] = sinf(0.37f * i + 1.f) * cosf(0.0011f * i * (i % 97) + 2.f);
            float *dx, *de; void* ws;
            hipMalloc(&dx, h.size() * 4); hipMalloc(&de, h.size() * 4);
            const size_t wsb = stof_hilbert_workspace_bytes(rows, n) + 256;
            hipMalloc(&ws, wsb);
            hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            const int st = stof_hilbert(dx, rows, n, de, nullptr, nullptr, ws, wsb, nullptr);
            hipDeviceSynchronize();
            std::vector<float> e(h.size());
            hipMemcpy(e.data(), de, h.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int r = 0; r < rows; ++r) {
                const std::vector<double> want = ref_envelope(h.data() + (size_t)r * n, n);
                for (int i = 0; i < n; ++i) { const double d = fabs(e[(size_t)r * n + i] - want[i]); if (!(d <= worst)) worst = d; }
            }
            printf("n=%d rows=%d status=%d max err %.3g %s\n", n, rows, st, worst, worst < 1e-5 ? "ok" : "MISMATCH");
            bad += !(worst < 1e-5);
            hipFree(dx); hipFree(de); hipFree(ws);
        }
    }
    return bad;
}
