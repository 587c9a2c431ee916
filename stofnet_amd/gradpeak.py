"""GradPeak / toa_detect / grad_peak_detect on the gfx950 kernels
(mirrors models/gradpeak.py:8-133)."""
import math

import numpy as np
import torch

from . import _lib
from .hilbert import hilbert_envelope

_MOMENT_SLOTS = 64     # STOF_MOMENT_SLOTS (include/stofnet_amd.h)
_CAP = 32          # echoes per row kept by the first pass; rows with more trigger a re-run
# Waveforms in (toa_detect), row lengths the fused kernels take: up to this many rows the envelope never leaves the chip
# (explicit threshold: ONE launch, stof_toa_detect; default threshold: stof_toa_moments, then the row kernel on the
# envelope it kept); larger batches run the envelope kernel and the row kernels, which are faster there -- the fused
# kernels hold a pair of rows per two waves, eight pairs per CU, and every extra round of pairs costs a full
# transform + streaming latency.  Measured on the MI355X per batch size: tools/time_gradpeak_paths.py ->
# profiles/r04_gradpeak_paths.json ([2048, 2000]: 67 vs 82 us per call, [4096, 2000]: 91 vs 88, [8192, 2000]: 157 vs 122,
# [32768, 2000]: 493 vs 321).  r4: the limit includes 4,096 rows (BASELINE config C2's batch): equal time there, and the
# envelope stays on the chip (1.1x the algorithmic HBM bytes instead of 3.1x).
_ONE_LAUNCH_MAX_ROWS = 4096


def gaussian_kernel_1d(sigma: float, num_sigmas: float = 3.) -> torch.Tensor:
    """models/gradpeak.py:71-76.  float64 taps; sigma enters as float32 exactly as
    torch.distributions.Normal(loc=0, scale=sigma) holds it."""
    if not sigma > 0:
        raise ValueError(f'Expected parameter scale of distribution Normal to be > 0, but found {sigma}')
    radius = int(num_sigmas * sigma) + 1
    support = np.arange(-radius, radius + 1, dtype=np.float64)
    s32 = np.float32(sigma)
    var = np.float64(np.float32(s32 * s32))
    log_scale = np.float64(np.log(s32, dtype=np.float32))
    k = np.exp(-(support ** 2) / (2.0 * var) - log_scale - math.log(math.sqrt(2.0 * math.pi)))
    return torch.from_numpy(k * (1.0 / k.sum()))


_TAPS = {}


def _taps_on(device, grad_step):
    """Gaussian taps for sigma = (2 g - 1) / 6 as a float32 device tensor (cached per device and step)."""
    key = (str(device), int(grad_step))
    if key not in _TAPS:
        _TAPS[key] = gaussian_kernel_1d((grad_step * 2 - 1) / 6).to(device, torch.float32)
    return _TAPS[key]


def _moment_reduce(stats, group):
    """Q7 across ranks, OPT-IN: the default threshold is the std of the WHOLE batch (models/gradpeak.py:18), so a batch
    whose rows are sharded over ranks sums its three moments (sum, sum of squares, count) over `group` before the
    threshold is formed.  The reference has no collective here, so nothing is reduced unless the caller says the batch
    is sharded (`group=` a process group, or `sharded=True` for the default group): a rank that calls GradPeak alone,
    or ranks that all hold the SAME unsharded rows (main.evaluate under torchrun), keep local moments like the
    reference.  With the reduction on, every rank of the group must make the same sequence of calls (lock-step)."""
    if group is None:
        return stats
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError('GradPeak: a sharded batch (group= / sharded=True) needs an initialised torch.distributed')
    pg = None if group is True else group
    import os
    if dist.get_world_size(pg) > 1 or os.environ.get('STOF_FORCE_COLLECTIVES') == '1':     # the switch: tests/test_rccl_one_rank.py
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=pg)
    return stats


def _detect(frame_or_env, is_frame, grad_step, threshold, ival, echo_max, group=None):
    """Shared driver of grad_peak_detect (envelope in) and toa_detect (waveform in): returns what the reference's
    grad_peak_detect + the echo_max block of toa_detect return.  One host read (flags) per call."""
    data = frame_or_env
    _lib.require_device(data, 'data')
    x = data.detach().contiguous().float()
    n, L = x.shape
    taps = _taps_on(x.device, grad_step)
    radius = (taps.numel() - 1) // 2
    lib = _lib.lib()
    stream = _lib.stream_ptr(x.device)
    emax = int(echo_max) if (echo_max is not None and echo_max != float('inf') and echo_max >= 1) else 0
    fused_ok = bool(is_frame and lib.stof_toa_detect_fused_ok(L, radius))
    one_launch = fused_ok and n <= _ONE_LAUNCH_MAX_ROWS
    fused = one_launch and threshold is not None
    env = None
    th_dev = None
    with torch.cuda.device(x.device):
        if not fused:
            if threshold is None:
                # Q7: (unbiased std of the WHOLE batch tensor) ** 16 * 1.2e13 (models/gradpeak.py:18), formed on the device
                stats = torch.tensor([0.0, 0.0, float(n * L)], dtype=torch.float64, device=x.device)
                if one_launch:
                    # one launch from the waveforms: envelope (kept for the detection below) and the moments of its
                    # smoothed gradient without the envelope being read back
                    env = torch.empty_like(x)
                    partials = torch.empty(_MOMENT_SLOTS * 16, dtype=torch.float64, device=x.device)
                    _lib.check(lib.stof_toa_moments(_lib.ptr(x), n, L, int(grad_step), _lib.ptr(taps), radius, _lib.ptr(env),
                                                    _lib.ptr(partials), _lib.ptr(stats), stream), 'stof_toa_moments')
                else:
                    env = hilbert_envelope(x, keep_cached=True) if is_frame else x
                    _lib.check(lib.stof_gradpeak_moments(_lib.ptr(env), n, L, int(grad_step), _lib.ptr(taps), radius,
                                                         _lib.ptr(stats), stream), 'stof_gradpeak_moments')
                _moment_reduce(stats, group)
                th_dev = torch.empty(1, dtype=torch.float32, device=x.device)
                _lib.check(lib.stof_gradpeak_threshold(_lib.ptr(stats), _lib.ptr(th_dev), stream), 'stof_gradpeak_threshold')
            else:
                env = hilbert_envelope(x, keep_cached=True) if is_frame else x

        def run(cap):
            echoes = torch.empty((n, cap, 3), dtype=torch.float32, device=x.device)
            reduced = torch.empty((n, emax, 3), dtype=torch.float32, device=x.device) if emax else None
            counts = torch.empty((n,), dtype=torch.int32, device=x.device)
            flags = torch.empty((2,), dtype=torch.int32, device=x.device)
            th = float(threshold) if threshold is not None else 0.0
            if fused:
                code = lib.stof_toa_detect(_lib.ptr(x), n, L, int(grad_step), _lib.ptr(taps), radius, th, ival[0], ival[1],
                                           emax, _lib.ptr(echoes), cap, _lib.ptr(reduced), _lib.ptr(counts),
                                           _lib.ptr(flags), None, stream)
            else:
                code = lib.stof_grad_peak_detect(_lib.ptr(env), n, L, int(grad_step), _lib.ptr(taps), radius, th,
                                                 _lib.ptr(th_dev), ival[0], ival[1], emax, _lib.ptr(echoes), cap,
                                                 _lib.ptr(reduced), _lib.ptr(counts), _lib.ptr(flags), stream)
            _lib.check(code, 'stof_toa_detect' if fused else 'stof_grad_peak_detect')
            return echoes, reduced, flags

        echoes, reduced, flags = run(_CAP)
        q9, kmax = (int(v) for v in flags.cpu()) if n else (0, 0)        # the one host read (reference: :35-66 python loop)
        if kmax > _CAP and not q9:
            echoes, reduced, flags = run(kmax)
    if q9:
        # Q9 (models/gradpeak.py:54-55): the reference returns an empty [3, 0] tensor for the whole batch
        return torch.tensor([[], [], []])
    if kmax == 0:
        return torch.zeros((n, 0), dtype=x.dtype, device=x.device)       # shape the reference builds (:66)
    if echo_max is not None and kmax > echo_max:                          # toa_detect's reduction (:107-114)
        out = reduced if emax else echoes[:, :0]
    else:
        out = echoes[:, :kmax]
    return out.to(data.dtype)


def grad_peak_detect(data, grad_step: int = None, threshold: float = None, ival_smin: int = None,
                     ival_smax: int = None, group=None, sharded=False):
    """models/gradpeak.py:8-68 -> [N, Kmax, 3] = (onset, peak, amplitude), zero padded.  `group` / `sharded=True`:
    the rows are one shard of a batch spread over that process group (default group) -- see _moment_reduce; by default
    the moments behind the default threshold are local, as in the reference."""
    grad_step = grad_step if grad_step is not None else 2
    if ival_smin is not None and ival_smax is not None:
        ival = (int(ival_smin), int(ival_smax))
    else:
        ival = (grad_step // 2, grad_step * 3)
    return _detect(data, False, grad_step, threshold, ival, None, group if group is not None else (True if sharded else None))


def toa_detect(frame, threshold=None, rescale_factor=1, echo_max=float('inf'), group=None, sharded=False):
    """models/gradpeak.py:99-116: Hilbert envelope -> grad_peak_detect -> top-`echo_max` echoes by amplitude in time
    order.  With an explicit threshold and a row length the fused kernel supports this is ONE launch (stof_toa_detect);
    otherwise envelope kernel + (moments + threshold +) detection kernel.  Nothing here runs on ATen."""
    return _detect(frame, True, rescale_factor // 6 * 5, threshold, (int(rescale_factor), 50 * int(rescale_factor)),
                   echo_max, group if group is not None else (True if sharded else None))


class GradPeak(torch.nn.Module):
    def __init__(self, threshold=None, rescale_factor=1, echo_max=float('inf'), onset_opt=False, group=None, sharded=False):
        """Reference signature (models/gradpeak.py:120) + `group` / `sharded` (new, default off): set when every rank
        feeds a different shard of one batch and the default threshold must still be the whole batch's."""
        super().__init__()
        self.threshold = threshold
        self.onset_opt = onset_opt
        self._fun = lambda x: toa_detect(x, threshold=threshold, rescale_factor=rescale_factor, echo_max=echo_max,
                                         group=group, sharded=sharded)

    def forward(self, x):
        echoes = self._fun(x.squeeze(1))
        return echoes[..., 1] if not self.onset_opt else echoes[..., 0]
