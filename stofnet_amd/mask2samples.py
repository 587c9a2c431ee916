"""mask2coords on the gfx950 picker kernels (mirrors utils/mask2samples.py:26-34,81-148).

`get_maxima_positions` and the padded scatter run on the device; the only host
sync is the one the reference has too (`int(max(counts))`, :93).  The rarely used
`echo_max` reduction (:105-110,117-132) is a few tensor ops on the tiny [N,Kmax]
result, as in the reference.
"""
import torch

from . import _lib

_IDX_CAP = 32            # detections per row kept by the first pass; larger rows trigger a re-run


def _pick(scores: torch.Tensor, window_size: int, threshold, cap: int, counts=None, idx=None):
    _lib.require_device(scores, 'scores')
    if scores.dim() != 3 or scores.shape[1] != 1:
        raise RuntimeError('mask2coords expects scores of shape [N, 1, M] (utils/mask2samples.py:32)')
    s = scores.detach().contiguous().float()
    n, _, m = s.shape
    if counts is None:
        counts = torch.empty((n,), dtype=torch.int32, device=s.device)
    if idx is None:
        idx = torch.empty((n, max(cap, 1)), dtype=torch.int32, device=s.device)
    if (counts.dtype != torch.int32 or idx.dtype != torch.int32 or counts.shape != (n,) or idx.dim() != 2
            or idx.shape[0] != n or not counts.is_contiguous() or not idx.is_contiguous()):
        raise ValueError('counts must be int32 [N], idx int32 [N, cap], both contiguous')
    has_th = 1 if threshold else 0                     # Q4: `if threshold:` (utils/mask2samples.py:16)
    with torch.cuda.device(s.device):
        _lib.check(_lib.lib().stof_pick_maxima(_lib.ptr(s), n, m, int(window_size), has_th,
                                               float(threshold) if threshold else 0.0,
                                               _lib.ptr(counts), _lib.ptr(idx), idx.shape[1],
                                               _lib.stream_ptr(s.device)), 'stof_pick_maxima')
    return s, counts, idx


def _pick_all(scores, window_size, threshold):
    s, counts, idx = _pick(scores, window_size, threshold, _IDX_CAP)
    kmax = int(counts.max()) if counts.numel() else 0   # host sync, as utils/mask2samples.py:93
    if kmax > idx.shape[1]:
        s, counts, idx = _pick(scores, window_size, threshold, kmax)
    return s, counts, idx, kmax


def pick_async(scores, window_size, threshold=None, cap=_IDX_CAP, counts=None, idx=None):
    """get_maxima_positions without the reference's host sync (utils/mask2samples.py:93): returns (counts[N] int32,
    idx[N, cap] int32) -- exact counts, the first `cap` detections of every row, entries beyond a row's count left as
    they were -- optionally written into caller-provided buffers.  For pipelines that keep the GPU queue full
    (bench.py C4); the caller checks counts.max() <= cap once at the end."""
    _, counts, idx = _pick(scores, window_size, threshold, cap, counts, idx)
    return counts, idx


def get_maxima_positions(scores, window_size, threshold=None):
    """int64 [K, 2] (row, time), row-major (utils/mask2samples.py:26-34)."""
    s, counts, idx, kmax = _pick_all(scores, window_size, threshold)
    if kmax == 0:
        return torch.zeros((0, 2), dtype=torch.long, device=s.device)
    ar = torch.arange(kmax, device=s.device)[None, :]
    mask = ar < counts[:, None]
    rows = torch.arange(s.shape[0], device=s.device)[:, None].expand(-1, kmax)[mask]
    return torch.stack([rows.long(), idx[:, :kmax][mask].long()], dim=1)


def onset_indices(scores, window_size, threshold=None):
    """(counts[N] int32, idx[N, Kmax] int32) -- the integer onset sample indices."""
    s, counts, idx, kmax = _pick_all(scores, window_size, threshold)
    return counts, idx[:, :kmax]


def mask2coords(scores, window_size, threshold=None, upsample_factor=1, echo_max=None):
    s, counts, idx, kmax = _pick_all(scores, window_size, threshold)
    n = s.shape[0]
    if kmax == 0:                                        # utils/mask2samples.py:87-88
        return torch.zeros((n, s.shape[1], 1), device=s.device)
    coords = torch.empty((n, kmax), dtype=torch.float32, device=s.device)
    reduce = bool(echo_max) and echo_max < kmax
    with torch.cuda.device(s.device):
        _lib.check(_lib.lib().stof_indices_to_coords(_lib.ptr(counts), _lib.ptr(idx), idx.shape[1], n, kmax,
                                                     1.0 if reduce else float(upsample_factor),
                                                     _lib.ptr(coords), _lib.stream_ptr(s.device)),
                   'stof_indices_to_coords')
    if reduce:                                           # :105-107 -> reduce_echoes :117-132
        amplitudes = get_amplitudes(s, coords)
        coords = reduce_echoes(torch.dstack([coords, amplitudes]), echo_max=echo_max)[..., 0]
        coords = coords / upsample_factor
    elif echo_max and echo_max > kmax:                   # :108-110
        pad = torch.zeros(n, int(echo_max) - kmax, device=coords.device, dtype=coords.dtype)
        coords = torch.cat([coords, pad], dim=-1)
    return coords


def reduce_echoes(samples_and_amps, echo_max=100):
    echo_num = samples_and_amps.shape[1]
    channel_num = samples_and_amps.shape[-1]
    echoes = samples_and_amps
    if echo_num > echo_max:
        order = torch.argsort(samples_and_amps[..., 1], descending=True, dim=1)
        echoes = torch.gather(samples_and_amps, 1, order[..., None].repeat(1, 1, channel_num))[:, :echo_max]
        order = torch.argsort(echoes[..., 0], descending=False, dim=1)
        echoes = torch.gather(echoes, 1, order[..., None].repeat(1, 1, channel_num))
    return echoes


def get_amplitudes(frames, samples):
    return torch.gather(frames.squeeze(), -1, torch.round(samples).long())


def coords2mask(samples, ref):
    """utils/mask2samples.py:139-148 (training-loss helper; tiny scatter, left to torch)."""
    empty_mask = torch.zeros_like(ref)
    samples[samples < 0] = 0
    empty_mask.scatter_(2, samples, 1)
    empty_mask[..., :1] = 0
    return empty_mask
