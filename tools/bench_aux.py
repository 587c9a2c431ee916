"""Standalone HBM-bound kernels against the HBM roofline (MI355X: 8 TB/s spec, ~6.3 TB/s achievable).
Prints one JSON line per kernel: algorithmic GB/s = bytes the op must move / time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stofnet_amd import synth
from stofnet_amd import SampleShuffle1D, mask2coords
from stofnet_amd.hilbert import hilbert_envelope
from stofnet_amd.mask2samples import onset_indices
from stofnet_amd.gradpeak import toa_detect

dev = torch.device('cuda:0')
PEAK = 8000.0


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e-3


def report(name, nbytes, sec, **kw):
    print(json.dumps({'kernel': name, 'ms': round(sec * 1e3, 4), 'GB/s': round(nbytes / sec / 1e9, 1),
                      'frac_of_8TB/s': round(nbytes / sec / 1e9 / PEAK, 3), **kw}), flush=True)


N = 4096
ONLY = sys.argv[1] if len(sys.argv) > 1 else ''
for r in (() if ONLY else (4, 10, 20)):
    x = torch.randn(N, r, 2000, device=dev)
    shuf = SampleShuffle1D(r)
    report(f'sample_shuffle r={r} [{N},{r},2000]', 2 * x.numel() * 4, timeit(lambda: shuf(x)))
if not ONLY:
    x = torch.randn(N, 64, 500, device=dev)
    shuf = SampleShuffle1D(4)
    report('sample_shuffle r=4 C=16 [4096,64,500] (EDSR shape)', 2 * x.numel() * 4, timeit(lambda: shuf(x)))
for M in (() if ONLY else (8000, 20000, 40000)):
    y = torch.randn(N, 1, M, device=dev)
    report(f'pick_maxima argmax [{N},1,{M}]', y.numel() * 4, timeit(lambda: onset_indices(y, 20, None)), note='includes the Kmax host sync')
    report(f'pick_maxima th=2.5 [{N},1,{M}]', y.numel() * 4, timeit(lambda: mask2coords(y, 20, 2.5, 4)), note='includes host sync + scatter')
for n in (1536, 2000, 8000, 20000):
    rows = N if n <= 8000 else 1024
    x = torch.from_numpy(synth.synth_randn(rows, n, seed=1)).to(dev)[:, 0]
    report(f'hilbert envelope [{rows},{n}]', 2 * x.numel() * 4, timeit(lambda: hilbert_envelope(x)))
x = torch.from_numpy(synth.synth_echo(N, 2000, seed=3, noise=0.01)).to(dev)[:, 0]
report(f'gradpeak toa_detect [{N},2000] th=1e-3 rf=10', 2 * x.numel() * 4, timeit(lambda: toa_detect(x, 1e-3, 10), reps=5), note='hilbert + gradient + pairing + host syncs')
