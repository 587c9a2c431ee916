"""CPU restatement (numpy) of the onset pickers and the Hilbert envelope
(TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).  Parity status: pinned by
tests/test_oracle_golden.py against golden vectors captured from the reference.

Row-wise plain loops: these are checkers for small cases, not fast paths.
"""
from __future__ import annotations

import math
import numpy as np


# --------------------------------------------------------------------------
# mask2coords  (utils/mask2samples.py:5-34, 81-114)
# --------------------------------------------------------------------------
def nms_window(window_size: int) -> int:
    """Q3: utils/mask2samples.py:7 -- even windows grow by one (20 -> 21)."""
    return window_size // 2 * 2 + 1


def maxima_positions(scores: np.ndarray, window_size: int, threshold=None) -> np.ndarray:
    """utils/mask2samples.py:26-34 on scores[N, 1, M] -> int64 [K, 2] (row, time), row-major.

    A sample survives NMS iff it equals the max of its odd window (implicit -inf
    padding, :8-9); survivors keep their value, the rest become 0.  Then
    (Q4, :16) a truthy threshold zeroes everything below it, otherwise everything
    below the per-row max of the NMS output is zeroed (ties survive, Q5).  The
    detections are the non-zero entries (:32).
    """
    s = np.asarray(scores, dtype=np.float32)
    assert s.ndim == 3 and s.shape[1] == 1, "mask2coords needs [N,1,M] (squeeze(1), :32)"
    n, _, m = s.shape
    half = (nms_window(window_size) - 1) // 2
    out = []
    for row in range(n):
        v = s[row, 0]
        keep = np.zeros(m, dtype=np.float32)
        for i in range(m):
            lo, hi = max(0, i - half), min(m, i + half + 1)
            if v[i] == v[lo:hi].max():
                keep[i] = v[i]
        if threshold:
            keep[keep < np.float32(threshold)] = 0
        else:
            keep[keep < keep.max()] = 0
        for i in np.nonzero(keep)[0]:
            out.append((row, int(i)))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def mask2coords(scores, window_size, threshold=None, upsample_factor=1, echo_max=None) -> np.ndarray:
    """utils/mask2samples.py:81-114 -> float32 [N, Kmax] zero padded, in input-sample
    units (index / upsample_factor); shape [N, 1, 1] zeros when nothing is found (:87-88)."""
    s = np.asarray(scores, dtype=np.float32)
    idx = maxima_positions(s, window_size, threshold)
    n = s.shape[0]
    if idx.size == 0:
        return np.zeros((n, s.shape[1], 1), dtype=np.float32)
    counts = np.bincount(idx[:, 0], minlength=n)
    kmax = int(counts.max())
    coords = np.zeros((n, kmax), dtype=np.float32)
    fill = np.zeros(n, dtype=np.int64)
    for row, t in idx:
        coords[row, fill[row]] = np.float32(t)
        fill[row] += 1
    if echo_max and echo_max < kmax:                                     # :105-107, :117-132
        sq = s.reshape(n, -1) if n > 1 else s.reshape(1, -1)
        amps = np.take_along_axis(sq, np.round(coords).astype(np.int64), axis=-1)
        order = np.argsort(-amps, axis=1, kind='stable')[:, :echo_max]
        sel = np.take_along_axis(coords, order, axis=1)
        coords = np.sort(sel, axis=1)
    elif echo_max and echo_max > kmax:                                   # :108-110
        coords = np.concatenate([coords, np.zeros((n, int(echo_max) - kmax), np.float32)], axis=1)
    return (coords / np.float32(upsample_factor)).astype(np.float32)    # :112


def coords2mask(samples: np.ndarray, ref_shape) -> np.ndarray:
    """utils/mask2samples.py:139-148: one-hot scatter along the last dim, negative
    indices clamped to 0, index 0 cleared (placeholder for 'no echo')."""
    mask = np.zeros(ref_shape, dtype=np.float32)
    s = np.maximum(np.asarray(samples, dtype=np.int64), 0)
    n = mask.shape[0]
    for row in range(n):
        for c in range(mask.shape[1]):
            mask[row, c, s[row, c].reshape(-1)] = 1
    mask[..., :1] = 0
    return mask


# --------------------------------------------------------------------------
# Hilbert  (utils/hilbert.py:5-21)
# --------------------------------------------------------------------------
def hilbert_filter(n: int) -> np.ndarray:
    """Q6: bins 1 .. n//2-1 doubled, bins n//2+1 .. zeroed, bin 0 and bin n//2 kept
    (utils/hilbert.py:13-17).  For odd n bin n//2 is *not* doubled, unlike scipy."""
    h = np.zeros(n, dtype=np.float64)
    h[0] = 1.0
    h[1:n // 2] = 2.0
    h[n // 2] = 1.0
    return h


def hilbert_transform(y: np.ndarray, dtype=np.float64) -> np.ndarray:
    """Analytic signal along the last dim.  Computed in float64 by default (ground
    truth); the reference computes in float32/complex64 and agrees to ~1e-6."""
    y = np.asarray(y).astype(dtype)
    n = y.shape[-1]
    f = np.fft.fft(y, axis=-1)
    v = np.fft.ifft(f * hilbert_filter(n), axis=-1)
    return v


def hilbert_envelope(y: np.ndarray) -> np.ndarray:
    return np.abs(hilbert_transform(y)).astype(np.float32)


# --------------------------------------------------------------------------
# GradPeak  (models/gradpeak.py:8-133)
# --------------------------------------------------------------------------
class GradPeakDegenerate(RuntimeError):
    """Q9: models/gradpeak.py:54-55 returns an empty [3,0] tensor for the whole batch
    when a row has edges but no surviving candidate; downstream indexing then
    fails.  Pinned as 'raises'."""


def gaussian_kernel_1d(sigma: float, num_sigmas: float = 3.0) -> np.ndarray:
    """models/gradpeak.py:71-76: taps in float64 from the normal log-pdf, normalised."""
    radius = int(num_sigmas * sigma) + 1
    support = np.arange(-radius, radius + 1, dtype=np.float64)
    # torch.distributions.Normal(loc=0, scale=sigma) holds scale as a float32 tensor
    # (default dtype): sigma, sigma**2 and log(sigma) are float32-rounded, the
    # support is float64, so the expression promotes to float64.
    s32 = np.float32(sigma)
    var = np.float64(np.float32(s32 * s32))
    log_scale = np.float64(np.log(s32, dtype=np.float32))
    logp = -(support ** 2) / (2.0 * var) - log_scale - math.log(math.sqrt(2.0 * math.pi))
    k = np.exp(logp)
    return k * (1.0 / k.sum())


def gradient_uniform(x: np.ndarray, spacing: float) -> np.ndarray:
    """torch.gradient(x, spacing=g, dim=-1) (models/gradpeak.py:14): central
    differences /(2g) inside, one-sided /g at both edges; float32."""
    x = np.asarray(x, dtype=np.float32)
    g = np.empty_like(x)
    sp = np.float32(spacing)
    g[..., 1:-1] = (x[..., 2:] - x[..., :-2]) / (np.float32(2.0) * sp)
    g[..., 0] = (x[..., 1] - x[..., 0]) / sp
    g[..., -1] = (x[..., -1] - x[..., -2]) / sp
    return g


def gaussian_filter_1d(x: np.ndarray, sigma: float) -> np.ndarray:
    """models/gradpeak.py:89-96: zero-padded 'same' correlation with the float32-cast taps."""
    k = gaussian_kernel_1d(sigma).astype(np.float32)
    pad = len(k) // 2
    xp = np.pad(np.asarray(x, np.float32), [(0, 0)] * (x.ndim - 1) + [(pad, pad)])
    out = np.zeros(x.shape, dtype=np.float32)
    for j, kj in enumerate(k):
        out += kj * xp[..., j:j + x.shape[-1]]
    return out


def smoothed_gradient(env: np.ndarray, grad_step: int) -> np.ndarray:
    return gaussian_filter_1d(gradient_uniform(env, grad_step), (grad_step * 2 - 1) / 6)


def default_threshold(grad: np.ndarray) -> np.float32:
    """Q7: (std over the WHOLE batch tensor, unbiased) ** 16 * 1.2e13 (models/gradpeak.py:18)."""
    std = np.float32(np.std(grad.astype(np.float64), ddof=1))
    return np.float32(np.float32(std ** np.float32(16)) * np.float32(1.2e13))


def rising_edges(flags: np.ndarray) -> np.ndarray:
    """diff(int(flags)) == 1 -> index of the last-false sample (models/gradpeak.py:27-30)."""
    f = flags.astype(np.int32)
    return np.nonzero(np.diff(f) == 1)[0]


def pair_row(ap: np.ndarray, am: np.ndarray, ival_min, ival_max):
    """models/gradpeak.py:42-60 for one row: each falling-slope edge `am` takes the
    nearest preceding (<=) rising-slope edge `ap` (first edge if none precedes,
    Q8 sentinel 2**32); keep ival_min < am-ap < ival_max; per distinct ap keep
    the first am.  Returns (onsets, peaks) or None if nothing survives."""
    cands = []
    for m in am:
        prev = ap[ap <= m]
        a = prev.max() if prev.size else ap[0]
        cands.append((int(a), int(m)))
    cands = [(a, m) for a, m in cands if ival_min < (m - a) < ival_max]
    if not cands:
        return None
    onsets, peaks, seen = [], [], set()
    for a, m in cands:
        if a not in seen:
            seen.add(a)
            onsets.append(a)
            peaks.append(m)
    order = np.argsort(onsets, kind='stable')
    return np.asarray(onsets)[order], np.asarray(peaks)[order]


def grad_peak_detect(env: np.ndarray, grad_step=None, threshold=None, ival_smin=None, ival_smax=None,
                     grad_override: np.ndarray | None = None) -> np.ndarray:
    """models/gradpeak.py:8-68 -> float32 [N, Kmax, 3] = (onset, peak, env[peak]) zero padded."""
    env = np.asarray(env, np.float32)
    grad_step = grad_step if grad_step is not None else 2
    grad = smoothed_gradient(env, grad_step) if grad_override is None else grad_override
    th_pos = np.float32(threshold) if threshold is not None else default_threshold(grad)
    th_neg = -th_pos / np.float32(4)
    if ival_smin is not None and ival_smax is not None:
        ival = (ival_smin, ival_smax)
    else:
        ival = (grad_step // 2, grad_step * 3)
    rows = []
    for i in range(env.shape[0]):
        ap = rising_edges(grad[i] > th_pos)
        am = rising_edges(grad[i] < th_neg)
        if ap.size == 0 or am.size == 0:
            rows.append(None)
            continue
        pr = pair_row(ap, am, ival[0], ival[1])
        if pr is None:
            raise GradPeakDegenerate("row %d has edges but no surviving candidate (models/gradpeak.py:54-55)" % i)
        rows.append(pr)
    kmax = max([len(r[0]) for r in rows if r is not None], default=0)
    out = np.zeros((env.shape[0], kmax, 3), dtype=np.float32)
    for i, r in enumerate(rows):
        if r is None:
            continue
        k = len(r[0])
        out[i, :k, 0] = r[0]
        out[i, :k, 1] = r[1]
        out[i, :k, 2] = env[i, r[1]]
    return out


def toa_detect(frame: np.ndarray, threshold=None, rescale_factor=1, echo_max=float('inf'),
               env: np.ndarray | None = None) -> np.ndarray:
    """models/gradpeak.py:99-116.  frame [N, L]."""
    env = hilbert_envelope(frame) if env is None else np.asarray(env, np.float32)
    echoes = grad_peak_detect(env, grad_step=rescale_factor // 6 * 5, ival_smin=rescale_factor,
                              ival_smax=50 * rescale_factor, threshold=threshold)
    if echoes.shape[1] > echo_max:
        k = int(echo_max)
        order = np.argsort(-echoes[..., 2], axis=1, kind='stable')[:, :k]
        sel = np.take_along_axis(echoes, order[..., None].repeat(3, -1), axis=1)
        order2 = np.argsort(sel[..., 1], axis=1, kind='stable')
        echoes = np.take_along_axis(sel, order2[..., None].repeat(3, -1), axis=1)
    return echoes


def gradpeak_forward(x: np.ndarray, threshold=None, rescale_factor=1, echo_max=float('inf'),
                     onset_opt=False, env=None) -> np.ndarray:
    """models/gradpeak.py:127-133.  x [N, 1, L] -> [N, K] (onset column if onset_opt else peak)."""
    e = toa_detect(np.asarray(x)[:, 0, :], threshold, rescale_factor, echo_max, env=env)
    return e[..., 0] if onset_opt else e[..., 1]


# --------------------------------------------------------------------------
# host metrics  (utils/metrics.py:9-41, utils/gaussian.py:4-7)
# --------------------------------------------------------------------------
def toa_rmse(gt: np.ndarray, es: np.ndarray, tol=1) -> np.ndarray:
    """utils/metrics.py:9-41 -> [N, 7] = (rmse, precision, recall, jaccard, tp, fp, fn)."""
    gt = np.asarray(gt, np.float32)
    es = np.asarray(es, np.float32)
    n = gt.shape[0]
    mes, tps, fps, fns = (np.zeros(n, np.float32) for _ in range(4))
    valid = lambda v: v[(v != 0) & np.isfinite(v)]
    for i in range(n):
        g, e = valid(gt[i].reshape(-1)), valid(es[i].reshape(-1))
        if g.size == 0 or e.size == 0:
            continue
        mins = ((g[:, None] - e[None, :]) ** 2).min(-1)
        hit = mins <= tol
        with np.errstate(invalid='ignore'):
            mes[i] = np.float32(np.mean(mins[hit])) ** np.float32(.5) if hit.any() else np.float32('nan')
        tps[i] = hit.sum()
        fns[i] = (~hit).sum()
        fps[i] = e.size - tps[i]
    with np.errstate(invalid='ignore', divide='ignore'):
        jac = tps / (fns + tps + fps) * 100
        pre = tps / (fps + tps) * 100
        rec = tps / (fns + tps) * 100
    return np.stack([mes, pre, rec, jac, tps, fps, fns]).T.astype(np.float32)


def gaussian_kernel(size, sigma=1.0) -> np.ndarray:
    """utils/gaussian.py:4-7."""
    x = np.linspace(-size // 2 + 1, size // 2, size)
    k = np.exp(-np.power(x / sigma, 2) / 2)
    return k / np.sum(k)


# --------------------------------------------------------------------------
# dataset-side RF preparation (datasets/chirp_dataset.py:80-91, utils/transforms.py:13)
# --------------------------------------------------------------------------
def iq2rf(iq_data: np.ndarray, fc: float, fs: float, rescale_factor=1, normalize=True) -> np.ndarray:
    """The same numpy/scipy calls the reference makes (the arithmetic lives in scipy.interpolate.interp1d):
    linear resampling of the complex IQ trace on endpoint-inclusive grids, up-mixing, real part, then
    NormalizeVol.  float64, one row at a time; returns float64 [N, int(len*rescale_factor)]."""
    from scipy.interpolate import interp1d
    iq = np.asarray(iq_data)
    out = []
    for row in iq:
        n = row.shape[0]
        x = np.linspace(0, n / fs, num=n, endpoint=True)
        t = np.linspace(0, n / fs, num=int(n * rescale_factor), endpoint=True)
        y = interp1d(x, row, axis=0)(t)
        rf = (y * np.exp(2j * np.pi * fc * t)).real
        out.append(rf / np.abs(rf).max() if normalize else rf)
    return np.stack(out)
