"""mask2coords on the gfx950 picker kernels (mirrors utils/mask2samples.py:26-34,81-148).

`get_maxima_positions`, the padded scatter and the `echo_max` reduction run on the
device; the only host sync is the one the reference has too (`int(max(counts))`, :93).
"""
import torch

from . import _lib

_IDX_CAP = 32            # detections per row kept by the first pass of arg-max mode (Kmax = 1 barring ties)
_TH_CAP = 128            # ... of threshold mode, where Kmax is data dependent (PALA, th = .015: mean 46, max 79 per row)
_last_kmax = {}          # (threshold mode?) -> Kmax of the previous call: the next call's first pass is sized for it, so a
                         # steady stream of similar batches (main.py:320 in an evaluation loop) takes ONE pass per call


def _pick(scores: torch.Tensor, window_size: int, threshold, cap: int, counts=None, idx=None):
    _lib.require_device(scores, 'scores')
    if scores.dim() != 3:
        raise RuntimeError('expected scores of shape [N, C, M]')
    s = scores.detach().contiguous().float()
    n, m = s.shape[0] * s.shape[1], s.shape[2]          # NMS and thresholding are per (batch, channel) row
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
    mode = bool(threshold)
    cap = max(_TH_CAP if mode else _IDX_CAP, min(_last_kmax.get(mode, 0) * 5 // 4, 4096))
    s, counts, idx = _pick(scores, window_size, threshold, cap)
    kmax = int(counts.max()) if counts.numel() else 0   # host sync, as utils/mask2samples.py:93
    if kmax > idx.shape[1]:                             # a row with more detections than the first pass kept: once more
        s, counts, idx = _pick(scores, window_size, threshold, kmax)
    _last_kmax[mode] = kmax
    return s, counts, idx, kmax


def pick_async(scores, window_size, threshold=None, cap=_IDX_CAP, counts=None, idx=None):
    """get_maxima_positions without the reference's host sync (utils/mask2samples.py:93): returns (counts[N] int32,
    idx[N, cap] int32) -- exact counts, the first `cap` detections of every row, entries beyond a row's count left as
    they were -- optionally written into caller-provided buffers.  For pipelines that keep the GPU queue full
    (bench.py C4); the caller checks counts.max() <= cap once at the end."""
    _, counts, idx = _pick(scores, window_size, threshold, cap, counts, idx)
    return counts, idx


def get_maxima_positions(scores, window_size, threshold=None):
    """utils/mask2samples.py:26-34: int64 [K, 2] (row, time) for scores [N, 1, M] (the reference squeezes dim 1, :32),
    [K, 3] (batch, channel, time) for C > 1; row-major like torch.nonzero."""
    s, counts, idx, kmax = _pick_all(scores, window_size, threshold)
    c = s.shape[1]
    if kmax == 0:
        return torch.zeros((0, 2 if c == 1 else 3), dtype=torch.long, device=s.device)
    ar = torch.arange(kmax, device=s.device)[None, :]
    mask = ar < counts[:, None]
    rows = torch.arange(counts.shape[0], device=s.device)[:, None].expand(-1, kmax)[mask].long()
    t = idx[:, :kmax][mask].long()
    if c == 1:
        return torch.stack([rows, t], dim=1)
    return torch.stack([rows // c, rows % c, t], dim=1)


def mask2nested_list(scores, window_size, threshold=None, upsample_factor=1):
    """utils/mask2samples.py:37-51 ('caution: computationally expensive'): nested [batch][channel] lists of numpy
    arrays of detections / upsample_factor.  The detection runs on the picker kernel; the list building is the
    reference's host loop over the index tensor, including its behaviour for [N, 1, M] scores, where column 1 of the
    (squeezed) index tensor is the TIME and the inner loop therefore runs over time indices."""
    indices = get_maxima_positions(scores, window_size, threshold).cpu()
    nested_list = []
    for bidx in range(int(indices[:, 0].max()) + 1):
        helper_list = []
        for cidx in range(int(indices[:, 1].max()) + 1):
            samples = indices[(indices[:, 0] == bidx) & (indices[:, 1] == cidx), -1] / upsample_factor
            helper_list.append(samples.numpy())
        nested_list.append(helper_list)
    return nested_list


def batch_mask2coords(scores, window_size, threshold=None, upsample_factor=1):
    """utils/mask2samples.py:54-78 for scores [B, C, M] with C > 1: float coords [b_max, c_max, K] where b_max / c_max
    are the largest batch / channel index WITH a detection + 1 and -- as in the reference, whose fill index counts the
    occupied (batch, channel) pairs (:71, enumerate over torch.unique) -- the i-th occupied pair fills row i of the
    flattened [b_max * c_max, K] tensor, so pairs without detections shift later ones up."""
    indices = get_maxima_positions(scores, window_size, threshold)
    if indices.numel() == 0:                              # :58-59
        return torch.zeros((scores.shape[0], scores.shape[1], 1), device=scores.device)
    if indices.shape[1] != 3:
        raise IndexError('index 2 is out of bounds for dimension 1 with size 2')      # the reference's indices[:, 2] (:64)
    b_max = int(indices[:, 0].max()) + 1
    c_max = int(indices[:, 1].max()) + 1
    samples = indices[:, 2].float() / upsample_factor
    flat2d = indices[:, 0] * c_max + indices[:, 1]
    _, inverse, counts = torch.unique(flat2d, return_inverse=True, return_counts=True)
    kmax = int(counts.max())
    start = torch.cumsum(counts, 0) - counts              # first detection of every occupied pair (row-major order)
    within = torch.arange(indices.shape[0], device=indices.device) - start[inverse]
    coords = torch.zeros((b_max, c_max, kmax), device=scores.device)
    coords.view(-1)[inverse * kmax + within] = samples
    return coords


def onset_indices(scores, window_size, threshold=None):
    """(counts[N] int32, idx[N, Kmax] int32) -- the integer onset sample indices."""
    s, counts, idx, kmax = _pick_all(scores, window_size, threshold)
    return counts, idx[:, :kmax]


def mask2coords(scores, window_size, threshold=None, upsample_factor=1, echo_max=None):
    """utils/mask2samples.py:81-114 on the picker kernels; the one host sync is the reference's own (`int(max(counts))`)."""
    s, counts, idx, kmax = _pick_all(scores, window_size, threshold)
    if s.shape[1] != 1:
        raise RuntimeError('mask2coords expects scores of shape [N, 1, M] (utils/mask2samples.py:32)')
    n = s.shape[0]
    if kmax == 0:                                        # utils/mask2samples.py:87-88
        return torch.zeros((n, s.shape[1], 1), device=s.device)
    lib = _lib.lib()
    with torch.cuda.device(s.device):
        if echo_max and echo_max < kmax:                 # :105-107 -> reduce_echoes :117-132 on the device
            k = int(echo_max)
            coords = torch.empty((n, k), dtype=torch.float32, device=s.device)
            _lib.check(lib.stof_reduce_echoes(_lib.ptr(s), n, s.shape[2], _lib.ptr(counts), _lib.ptr(idx), idx.shape[1], kmax,
                                              k, float(upsample_factor), _lib.ptr(coords), _lib.stream_ptr(s.device)),
                       'stof_reduce_echoes')
            return coords
        width = int(echo_max) if (echo_max and echo_max > kmax) else kmax      # :108-110: zero padding up to echo_max
        coords = torch.empty((n, width), dtype=torch.float32, device=s.device)
        _lib.check(lib.stof_indices_to_coords(_lib.ptr(counts), _lib.ptr(idx), idx.shape[1], n, width, float(upsample_factor),
                                              _lib.ptr(coords), _lib.stream_ptr(s.device)), 'stof_indices_to_coords')
    return coords


def reduce_echoes(samples_and_amps, echo_max=100):
    """utils/mask2samples.py:117-132 for callers that hold a [N, K, C] tensor already (mask2coords itself uses the
    stof_reduce_echoes kernel); a few tensor ops on a tiny result."""
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
