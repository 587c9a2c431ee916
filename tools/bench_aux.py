"""Standalone HBM-bound kernels against the HBM roofline (MI355X: 8 TB/s spec, ~6.3 TB/s achievable).

    python tools/bench_aux.py [substring | --list | --case NAME]

Prints one JSON line per case: wall time per call (launch + the host syncs the reference has too) and the algorithmic
GB/s = bytes the op must move / time.  `--case NAME` runs exactly one case (tools/profile_aux.sh wraps each case in its
own rocprofv3 run, so that every dispatch of the run belongs to that case: kernel-only microseconds and FETCH / WRITE
bytes per case end up in profiles/r02_aux_*.json).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stofnet_amd import GradPeak, SampleShuffle1D, mask2coords, synth
from stofnet_amd.gradpeak import toa_detect
from stofnet_amd.hilbert import hilbert_envelope
from stofnet_amd.mask2samples import onset_indices, pick_async

dev = torch.device('cuda:0')
PEAK = 8000.0
N = 4096


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


def case_shuffle(r, c, w):
    x = torch.randn(N, r * c, w, device=dev)
    shuf = SampleShuffle1D(r)
    return (lambda: shuf(x)), 2 * x.numel() * 4, 'read + write of the tensor'


def case_pick(m, th):
    y = torch.randn(N, 1, m, device=dev)
    if th is None:
        return (lambda: pick_async(y, 20, None)), y.numel() * 4, 'one read of the map (no host sync: pick_async)'
    return (lambda: pick_async(y, 20, th)), y.numel() * 4, 'one read of the map (no host sync: pick_async)'


def case_pick_sync(m, th):
    y = torch.randn(N, 1, m, device=dev)
    return (lambda: mask2coords(y, 20, th, 4)), y.numel() * 4, 'mask2coords end to end incl. the reference\'s Kmax host sync + scatter'


def case_hilbert(rows, n):
    x = torch.from_numpy(synth.synth_randn(min(rows, 4096), n, seed=1)).to(dev)[:, 0].contiguous()
    if rows > 4096:
        x = x.repeat(rows // 4096, 1).contiguous()
    return (lambda: hilbert_envelope(x)), 2 * x.numel() * 4, 'row read + envelope write'


def case_gradpeak(rows, n, rf, th, chirp=False, long_rows=False):
    kw = dict(noise=0.0005, attack=300, tau=3000.0, carrier=0.001) if long_rows else dict(noise=0.01)
    x = torch.from_numpy(synth.synth_echo(min(rows, 4096), n, seed=3, **kw)).to(dev)
    if rows > 4096:
        x = x.repeat(rows // 4096, 1, 1).contiguous()
    if chirp:
        gp = GradPeak(threshold=th, rescale_factor=rf, echo_max=1, onset_opt=True)
        return (lambda: gp(x)), x.numel() * 4, 'row read (echoes are a few bytes per row); GradPeak module, chirp config, incl. the one host read'
    xs = x[:, 0].contiguous()
    return (lambda: toa_detect(xs, th, rf)), x.numel() * 4, 'row read (echoes are a few bytes per row); incl. the one host read'


CASES = {
    'shuffle_r4': lambda: case_shuffle(4, 1, 2000),
    'shuffle_r10': lambda: case_shuffle(10, 1, 2000),
    'shuffle_r20': lambda: case_shuffle(20, 1, 2000),
    'shuffle_r4_c16_edsr': lambda: case_shuffle(4, 16, 500),
    'pick_argmax_m8000': lambda: case_pick(8000, None),
    'pick_argmax_m20000': lambda: case_pick(20000, None),
    'pick_argmax_m40000': lambda: case_pick(40000, None),
    'pick_th2p5_m8000': lambda: case_pick(8000, 2.5),
    'pick_th2p5_m20000': lambda: case_pick(20000, 2.5),
    'pick_th2p5_m40000': lambda: case_pick(40000, 2.5),
    'mask2coords_argmax_m20000': lambda: case_pick_sync(20000, None),
    'hilbert_4096x1536': lambda: case_hilbert(4096, 1536),
    'hilbert_4096x2000': lambda: case_hilbert(4096, 2000),
    'hilbert_16384x2000': lambda: case_hilbert(16384, 2000),
    'hilbert_65536x2000': lambda: case_hilbert(65536, 2000),
    'hilbert_4096x2048': lambda: case_hilbert(4096, 2048),
    'hilbert_4096x4000': lambda: case_hilbert(4096, 4000),
    'hilbert_4096x4096': lambda: case_hilbert(4096, 4096),
    'hilbert_4096x6144': lambda: case_hilbert(4096, 6144),
    'hilbert_4096x8000': lambda: case_hilbert(4096, 8000),
    'hilbert_2048x15360': lambda: case_hilbert(2048, 15360),
    'hilbert_1024x20000': lambda: case_hilbert(1024, 20000),
    'hilbert_512x30720': lambda: case_hilbert(512, 30720),
    # (row counts up to gradpeak._ONE_LAUNCH_MAX_ROWS take the fused kernels, larger batches envelope kernel + row kernels)
    'gradpeak_th1em3_2048x2000_rf10': lambda: case_gradpeak(2048, 2000, 10, 1e-3),
    'gradpeak_th1em3_4096x2000_rf10': lambda: case_gradpeak(4096, 2000, 10, 1e-3),
    'gradpeak_th1em3_32768x2000_rf10': lambda: case_gradpeak(32768, 2000, 10, 1e-3),
    'gradpeak_chirp_4096x2000_rf10_th1em3': lambda: case_gradpeak(4096, 2000, 10, 1e-3, chirp=True),
    'gradpeak_default_th_2048x2000_rf10': lambda: case_gradpeak(2048, 2000, 10, None),
    'gradpeak_default_th_4096x2000_rf10': lambda: case_gradpeak(4096, 2000, 10, None),
    'gradpeak_unfused_4096x4000_rf20_th1em3': lambda: case_gradpeak(4096, 4000, 20, 1e-3),
    'gradpeak_long_512x30720_rf20_th1em4': lambda: case_gradpeak(512, 30720, 20, 1e-4, long_rows=True),
}


def run(name):
    fn, nbytes, what = CASES[name]()
    sec = timeit(fn)
    print(json.dumps({'case': name, 'ms': round(sec * 1e3, 4), 'algorithmic_bytes': nbytes, 'GB/s': round(nbytes / sec / 1e9, 1),
                      'frac_of_8TB/s': round(nbytes / sec / 1e9 / PEAK, 3), 'bytes_counted': what}), flush=True)


if __name__ == '__main__':
    arg = sys.argv[1:]
    if arg[:1] == ['--list']:
        print('\n'.join(CASES))
    elif arg[:1] == ['--case']:
        run(arg[1])
    else:
        for name in CASES:
            if not arg or arg[0] in name:
                run(name)
