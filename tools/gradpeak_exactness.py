#!/usr/bin/env python3
"""How exact are the GradPeak indices?  ('unquantified tolerance')

For every 1024-row golden case of the reference (tests/golden/f9_gradpeak_1024.npz: rf 10 / 20 x explicit / default
threshold) and the 4096-row case of test_gradpeak_many_rows_margin_gated_exactness this prints and stores

  * rows whose (onset, peak) indices differ from the REFERENCE's own fp32 result (the golden),
  * for each such row what the float64-exact pipeline decides (fp64 FFT, fp64 gradient and blur with the reference's
    fp32-cast taps, exact comparison): 'gpu' if the kernels agree with it, 'reference' if torch's fp32 does, 'neither',
  * rows where the GPU differs from the float64-exact result, and the same count for the reference's fp32 golden,
  * the smallest distance of the deciding sample from its threshold in those rows (units of max|gradient|).

Needs the GPU (runs the product kernels); uses oracle/ only as the checker.  Output: gpurun_out/r03_gradpeak_exactness.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pickers_oracle as po  # noqa: E402
from stofnet_amd import synth, toa_detect  # noqa: E402
import stofnet_amd.gradpeak as gp  # noqa: E402


def exact_pipeline(frame, rf, th):
    """float64 all the way: envelope, torch.gradient's formula, zero-padded blur with the reference's taps (which ARE
    float32 numbers: models/gradpeak.py:91 casts them), thresholds; then the reference's edge / pairing rules."""
    g = rf // 6 * 5
    env = np.abs(po.hilbert_transform(frame, np.float64))
    grad = np.empty_like(env)
    grad[:, 1:-1] = (env[:, 2:] - env[:, :-2]) / (2.0 * g)
    grad[:, 0] = (env[:, 1] - env[:, 0]) / g
    grad[:, -1] = (env[:, -1] - env[:, -2]) / g
    k = po.gaussian_kernel_1d((g * 2 - 1) / 6).astype(np.float32).astype(np.float64)
    pad = len(k) // 2
    xp = np.pad(grad, [(0, 0), (pad, pad)])
    sm = np.zeros_like(grad)
    for j, kj in enumerate(k):
        sm += kj * xp[:, j:j + grad.shape[1]]
    if th is None:
        std = np.std(sm, ddof=1)
        thp = std ** 16 * 1.2e13
    else:
        thp = float(np.float32(th))
    rows = []
    for i in range(frame.shape[0]):
        ap = po.rising_edges(sm[i] > thp)
        am = po.rising_edges(sm[i] < -thp / 4)
        if ap.size == 0 or am.size == 0:
            rows.append((np.zeros(0, int), np.zeros(0, int)))
            continue
        pr = po.pair_row(ap, am, rf, 50 * rf)
        rows.append(pr if pr is not None else (np.zeros(0, int), np.zeros(0, int)))
    return rows, sm, thp


def rows_of(arr):
    """[N, K, >=2] zero padded -> list of (onsets, peaks) with the padding stripped."""
    out = []
    for r in arr:
        keep = (r[:, 0] != 0) | (r[:, 1] != 0)
        out.append((r[keep, 0].astype(int), r[keep, 1].astype(int)))
    return out


def same(a, b):
    return len(a[0]) == len(b[0]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def margin(sm_row, thp):
    return float(min(np.abs(sm_row - thp).min(), np.abs(sm_row + thp / 4).min()))


def one_case(name, frame, rf, th, ref_idx):
    dev = torch.device('cuda:0')
    x = torch.from_numpy(frame).to(dev)
    got = toa_detect(x, threshold=th, rescale_factor=rf).cpu().numpy()
    g_rows = rows_of(got)
    exact, sm, thp = exact_pipeline(frame.astype(np.float64), rf, th)
    scale = float(np.abs(sm).max())
    rec = {'case': name, 'rows': int(frame.shape[0]), 'L': int(frame.shape[1]), 'rf': rf,
           'threshold': 'default (Q7)' if th is None else th, 'threshold_value_fp64': thp, 'max_abs_gradient': scale}
    gpu_vs_exact = [i for i in range(len(g_rows)) if not same(g_rows[i], exact[i])]
    rec['gpu_rows_differing_from_float64_exact'] = len(gpu_vs_exact)
    if ref_idx is not None:
        r_rows = rows_of(ref_idx)
        diff = [i for i in range(len(g_rows)) if not same(g_rows[i], r_rows[i])]
        ref_vs_exact = [i for i in range(len(r_rows)) if not same(r_rows[i], exact[i])]
        rec['gpu_rows_differing_from_reference_fp32'] = len(diff)
        rec['reference_fp32_rows_differing_from_float64_exact'] = len(ref_vs_exact)
        rec['differing_rows'] = [{'row': int(i),
                                  'float64_exact_agrees_with': ('gpu' if same(g_rows[i], exact[i]) else
                                                                'reference' if same(r_rows[i], exact[i]) else 'neither'),
                                  'closest_sample_to_a_threshold_rel': margin(sm[i], thp) / scale,
                                  'gpu': [g_rows[i][0].tolist(), g_rows[i][1].tolist()],
                                  'reference': [r_rows[i][0].tolist(), r_rows[i][1].tolist()]} for i in diff]
    rec['gpu_vs_exact_rows'] = [{'row': int(i), 'closest_sample_to_a_threshold_rel': margin(sm[i], thp) / scale} for i in gpu_vs_exact[:16]]
    print(json.dumps({k: v for k, v in rec.items() if k not in ('differing_rows', 'gpu_vs_exact_rows')}), flush=True)
    return rec


def main():
    golden = np.load(os.path.join(ROOT, 'tests', 'golden', 'f9_gradpeak_1024.npz'))
    out = {'what': __doc__.split('\n\n')[0], 'cases': []}
    for rf in (10, 20):
        L, seed = int(golden[f'L_rf{rf}']), int(golden[f'seed_rf{rf}'])
        frame = synth.synth_echo(1024, L, seed=seed, noise=0.01)[:, 0]
        for thn, th in (('1em3', 1e-3), ('none', None)):
            out['cases'].append(one_case(f'f9_gradpeak_1024 rf{rf} th{thn}', frame, rf, th, golden[f'idx_rf{rf}_th{thn}']))
    frame = synth.synth_echo(4096, 2000, seed=77, noise=0.01)[:, 0]
    out['cases'].append(one_case('4096 rows (seed 77) rf10 th1e-3, no reference golden', frame, 10, 1e-3, None))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r03_gradpeak_exactness.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
