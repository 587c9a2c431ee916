"""Deterministic synthetic RF waveforms (SURVEY.md §8d "Synthetic inputs").

The reference's chirp dataset is absent (`.MISSING_LARGE_BLOBS:2`), so parity
and throughput runs use these generators.  numpy's PCG64 `default_rng` stream
is stable across numpy versions, so a seed identifies the data on any box.
The waveforms follow what `datasets/chirp_dataset.py:112-122` hands the
network: an RF echo, max-abs normalised (`utils/transforms.py:13`).
"""
import numpy as np


def synth_echo(n_rows: int, length: int, seed: int, *, noise: float = 0.03,
               attack: int = 30, tau: float = 150.0, carrier: float = 0.02,
               return_onsets: bool = False):
    """Echo = linear attack * exp decay * sine carrier + Gaussian noise, |x|max = 1.

    onset ~ U[200, 0.85*length) (clipped to the row for short rows).
    Returns float32 [n_rows, 1, length] (and the integer onsets if asked).
    """
    rng = np.random.default_rng(seed)
    lo = min(200, max(1, length // 10))
    hi = max(lo + 1, int(0.85 * length))
    onsets = rng.integers(lo, hi, size=n_rows)
    t = np.arange(length, dtype=np.float64)[None, :]
    rel = t - onsets[:, None].astype(np.float64)
    env = np.clip(rel / attack, 0.0, 1.0) * np.exp(-np.maximum(rel - attack, 0.0) / tau)
    env[rel < 0] = 0.0
    phase = rng.uniform(0, 2 * np.pi, size=(n_rows, 1))
    sig = env * np.sin(2 * np.pi * carrier * rel + phase)
    sig = sig + noise * rng.standard_normal(size=sig.shape)
    sig /= np.abs(sig).max(axis=1, keepdims=True)
    out = sig.astype(np.float32)[:, None, :]
    return (out, onsets) if return_onsets else out


def synth_randn(n_rows: int, length: int, seed: int):
    """Pure Gaussian rows, max-abs normalised per row (throughput input shape)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(size=(n_rows, length))
    x /= np.abs(x).max(axis=1, keepdims=True)
    return x.astype(np.float32)[:, None, :]


def synth_conv_last(r: int, seed: int, num_features: int = 64, k: int = 3):
    """Seeded conv_last (r, 64, 3) weights/bias for the r=10/20 north-star shapes
    (all shipped checkpoints have r=4).  Same scale as
    PyTorch's default Conv1d init bound 1/sqrt(fan_in)."""
    rng = np.random.default_rng(seed)
    bound = 1.0 / np.sqrt(num_features * k)
    w = rng.uniform(-bound, bound, size=(r, num_features, k)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(r,)).astype(np.float32)
    return w, b


def synth_state_dict(r: int, seed: int, semi_global_scale: int = 80, num_features: int = 64):
    """Seeded-random full parameter set with the reference's names/shapes
    (`models/stofnet.py:23-31,88,94`), PyTorch-default-like uniform bounds."""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, co, ci, k):
        bound = 1.0 / np.sqrt(ci * k)
        sd[name + '.weight'] = rng.uniform(-bound, bound, size=(co, ci, k)).astype(np.float32)
        sd[name + '.bias'] = rng.uniform(-bound, bound, size=(co,)).astype(np.float32)

    conv('conv1', num_features, 1, 9)
    conv('conv_last', r, num_features, 3)
    if semi_global_scale != 1:
        fs = max(1, semi_global_scale // 10)
        conv('semi_global_block.contract_conv', fs * num_features, num_features, 5)
        conv('semi_global_block.expand_conv', num_features, fs * num_features, 5)
    for i in range(2, 13):
        conv(f'conv{i}', num_features, num_features, 7)
    return sd


def pala_frames(B: int, C: int, S: int, seed: int):
    """Synthetic PALA-like RF frame stack [B, C, S] (the reference's PALA loader yields [B, waves, C, S] and main.py:301
    flattens one wave to [B*C, 1, S]; the dataset itself is an absent submodule): a few micro-bubble echoes per frame,
    each arriving at a channel-dependent delay (hyperbolic move-out), long smooth pulses as after the x20 interpolation
    (rf_scale_factor 20), NormalizeVol over the whole frame (utils/transforms.py:13)."""
    rng = np.random.default_rng(seed)
    t = np.arange(S, dtype=np.float64)[None, None, :]
    ch = np.arange(C, dtype=np.float64)[None, :, None]
    out = np.zeros((B, C, S))
    for _ in range(3):
        depth = rng.uniform(0.15, 0.7, size=(B, 1, 1)) * S
        xpos = rng.uniform(0, C, size=(B, 1, 1))
        delay = np.sqrt(depth ** 2 + (40.0 * (ch - xpos)) ** 2)
        rel = t - delay
        env = np.clip(rel / 300.0, 0, 1) * np.exp(-np.maximum(rel - 300.0, 0) / 900.0)
        env[rel < 0] = 0
        out += rng.uniform(0.4, 1.0, size=(B, 1, 1)) * env * np.sin(2 * np.pi * 0.004 * rel + rng.uniform(0, 6.28, size=(B, 1, 1)))
    out += 0.0005 * rng.standard_normal(out.shape)
    out /= np.abs(out).max(axis=(1, 2), keepdims=True)
    return out.astype(np.float32)
