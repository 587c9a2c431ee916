"""CPU check of the device FFT building blocks (stofnet_amd/csrc/fft_small.h): the header is host/device neutral, so a
g++-built harness runs the very code the Hilbert / GradPeak kernels execute -- plan, two-level twiddle tables,
register butterflies (incl. the composite radix 8/16/25 ones), in-place DIF/DIT passes and the fused middle pass with
the Hilbert filter -- against the pinned oracle (oracle/pickers_oracle.py:hilbert_transform, float64)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import pickers_oracle as po

SRC = os.path.join(ROOT, 'tests', 'cpu_harness', 'fft_harness.cpp')


@pytest.fixture(scope='module')
def harness(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('fft') / 'fft_harness.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-fconstexpr-ops-limit=200000000', '-fconstexpr-loop-limit=10000000', '-shared', '-fPIC', '-Wno-unknown-pragmas', '-o', so, SRC], check=True)
    lib = ctypes.CDLL(so)
    lib.fft_plan.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.fft_analytic.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.fft_butterfly.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.fft_analytic_ct.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.fft_ct_table.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.fft_analytic_ct_plain.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.fft_ct_plan.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    return lib


def plan(lib, n):
    r = (ctypes.c_int * 16)()
    k = lib.fft_plan(n, r)
    return list(r[:k])


@pytest.mark.parametrize('R', [2, 3, 4, 5, 6, 8, 10, 15, 16, 20, 25])
def test_butterflies_match_numpy_dft(harness, R):
    rng = np.random.default_rng(R)
    z = (rng.standard_normal(R) + 1j * rng.standard_normal(R)).astype(np.complex64)
    buf = z.copy()
    assert harness.fft_butterfly(R, buf.ctypes.data)
    assert np.abs(buf - np.fft.fft(z.astype(np.complex128))).max() < 2e-6 * R


def test_plan_shapes(harness):
    for n in [2, 4, 6, 10, 16, 96, 1536, 2000, 2048, 4000, 8000, 15360, 16000, 19660, 20000, 20480, 30720, 40000]:
        p = plan(harness, n)
        if n == 19660:                                   # 983 is prime
            assert p == []
            continue
        assert p and int(np.prod(p)) == n and p[-1] in (2, 4) and all(r in (2, 3, 4, 5, 8, 16) for r in p), (n, p)
    assert plan(harness, 2000) == [5, 5, 5, 4, 4] and plan(harness, 1536) == [16, 8, 3, 4]
    assert plan(harness, 7) == [] and plan(harness, 2001) == [] and plan(harness, 14) == []


@pytest.mark.parametrize('n', [2, 4, 6, 8, 10, 12, 16, 20, 50, 96, 160, 1536, 2000, 2048, 2560, 4000, 8000, 20000])
@pytest.mark.parametrize('nthreads', [1, 64])
def test_analytic_signal_matches_oracle(harness, n, nthreads):
    rng = np.random.default_rng(n)
    x1 = rng.standard_normal(n)
    x2 = rng.standard_normal(n)
    x1 /= np.abs(x1).max()
    x2 /= np.abs(x2).max()
    z = (x1 + 1j * x2).astype(np.complex64)              # two real rows ride one complex transform
    buf = z.copy()
    assert harness.fft_analytic(n, buf.ctypes.data, nthreads)
    a1 = po.hilbert_transform(x1.astype(np.float32))     # float64 analytic signals of the fp32 rows
    a2 = po.hilbert_transform(x2.astype(np.float32))
    want = a1 + 1j * a2                                  # linearity: ifft(H fft(x1 + i x2))
    assert np.abs(buf - want).max() < 1e-5 * max(1.0, np.log2(n) / 8)
    # un-mixing used by the kernels: v1 = Im - x2, v2 = x1 - Re
    v1 = buf.imag - z.imag
    v2 = z.real - buf.real
    assert np.abs(np.hypot(z.real, v1) - np.abs(a1)).max() < 1e-5
    assert np.abs(np.hypot(z.imag, v2) - np.abs(a2)).max() < 1e-5


@pytest.mark.parametrize('n', [96, 1536, 2000, 2048, 4000, 4096, 8000])
@pytest.mark.parametrize('nthreads', [64, 128, 256])
def test_compile_time_plans_match_oracle(harness, n, nthreads):
    """analytic_ct<N, T>: constant strides / trip counts, full compile-time twiddle table (the kernels' fast path)."""
    rng = np.random.default_rng(n + 7)
    x1 = rng.standard_normal(n)
    x2 = rng.standard_normal(n)
    x1 /= np.abs(x1).max()
    x2 /= np.abs(x2).max()
    z = (x1 + 1j * x2).astype(np.complex64)
    buf = z.copy()
    assert harness.fft_analytic_ct(n, buf.ctypes.data, nthreads)
    want = po.hilbert_transform(x1.astype(np.float32)) + 1j * po.hilbert_transform(x2.astype(np.float32))
    assert np.abs(buf - want).max() < 1e-5 * max(1.0, np.log2(n) / 8)
    ref = z.copy()                                        # the run-time plan computes the same transform
    assert harness.fft_analytic(n, ref.ctypes.data, 1)
    assert np.abs(buf - ref).max() < 3e-6


@pytest.mark.parametrize('n', [1536, 2000])
def test_compile_time_twiddle_table_is_correctly_rounded(harness, n):
    out = np.zeros(2 * n, np.float32)
    k = harness.fft_ct_table(n, out.ctypes.data)
    assert k > 0
    want = np.exp(-2j * np.pi * np.arange(k) / n)
    got = out[0:2 * k:2].astype(np.float64) + 1j * out[1:2 * k:2]
    assert np.abs(got - want).max() < 6e-8                # half an ulp of fp32 near 1


def test_compile_time_plan_shapes(harness):
    """N = 16 (middle pass) x radices from {16, 10, 8, 6, 5, 4, 3, 2}; the table holds the first N / min radix powers."""
    def ct_plan(n):
        r = (ctypes.c_int * 16)()
        t = ctypes.c_int()
        k = harness.fft_ct_plan(n, r, ctypes.byref(t))
        return (list(r[:k]), t.value) if k >= 0 else None
    assert ct_plan(2000) == ([5, 5, 5], 400)
    assert ct_plan(1536) == ([16, 6], 256)
    assert ct_plan(4000) == ([10, 5, 5], 800)
    assert ct_plan(4096) == ([16, 16], 256)
    assert ct_plan(16) == ([], 1)
    assert ct_plan(2008) is None and ct_plan(112) is None and ct_plan(24) is None


@pytest.mark.parametrize('n', [2000, 8000, 20000])
def test_compile_time_plan_plain_layout_two_level_twiddles(harness, n):
    """CtOpt<PAD = false, TW2 = true>: the unpadded image and the two-level twiddle table used when the padded image of a
    row (20,000 values) would not fit LDS."""
    rng = np.random.default_rng(n + 11)
    x1 = rng.standard_normal(n)
    x2 = rng.standard_normal(n)
    x1 /= np.abs(x1).max()
    x2 /= np.abs(x2).max()
    z = (x1 + 1j * x2).astype(np.complex64)
    buf = z.copy()
    assert harness.fft_analytic_ct_plain(n, buf.ctypes.data)
    want = po.hilbert_transform(x1.astype(np.float32)) + 1j * po.hilbert_transform(x2.astype(np.float32))
    assert np.abs(buf - want).max() < 1e-5 * max(1.0, np.log2(n) / 8)
