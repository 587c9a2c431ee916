"""hilbert_transform / HilbertTransform on the gfx950 LDS-FFT kernel
(mirrors utils/hilbert.py:5-34)."""
import torch
import torch.nn as nn

from . import _lib


def _run(y: torch.Tensor, want_env: bool, want_complex: bool, keep_cached: bool = False):
    _lib.require_device(y, 'y')
    n = y.shape[-1]
    rows = y.numel() // max(n, 1)
    lib = _lib.lib()
    if y.dtype == torch.float64:
        # utils/hilbert.py:11: torch.fft.fft follows the input's dtype -- a float64 frame stays in complex128
        yc = y.detach().contiguous().reshape(rows, n)
        ws = torch.empty(max(lib.stof_hilbert_f64_workspace_bytes(rows, n), 16), dtype=torch.uint8, device=y.device)
        env = torch.empty_like(yc) if want_env else None
        re = torch.empty_like(yc) if want_complex else None
        im = torch.empty_like(yc) if want_complex else None
        with torch.cuda.device(y.device):
            _lib.check(lib.stof_hilbert_f64(_lib.ptr(yc), rows, n, _lib.ptr(env), _lib.ptr(re), _lib.ptr(im),
                                            _lib.ptr(ws), ws.numel(), _lib.stream_ptr(y.device)), 'stof_hilbert_f64')
        return env, re, im
    yc = y.detach().contiguous().float().reshape(rows, n)
    ws = torch.empty(max(lib.stof_hilbert_workspace_bytes(rows, n), 16), dtype=torch.uint8, device=y.device)
    env = torch.empty_like(yc) if want_env else None
    re = torch.empty_like(yc) if want_complex else None
    im = torch.empty_like(yc) if want_complex else None
    with torch.cuda.device(y.device):
        call = lib.stof_hilbert if keep_cached else lib.stof_hilbert_streamed
        _lib.check(call(_lib.ptr(yc), rows, n, _lib.ptr(env), _lib.ptr(re), _lib.ptr(im),
                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr(y.device)), 'stof_hilbert')
    return env, re, im


def hilbert_transform(y):
    """Analytic signal along the last dim, with the reference's bin rule (Q6): complex128 for a float64 input,
    complex64 otherwise (torch.fft.fft's dtype rule, utils/hilbert.py:11)."""
    _, re, im = _run(y, False, True)
    return torch.complex(re, im).reshape(y.shape)


def hilbert_envelope(y, keep_cached=False):
    """abs(hilbert_transform(y)) without materialising the complex signal.  keep_cached: a kernel of this package reads
    the envelope next (GradPeak); otherwise it is written with non-temporal stores (stof_hilbert_streamed)."""
    env, _, _ = _run(y, True, False, keep_cached)
    return env.reshape(y.shape)


class HilbertTransform(nn.Module):
    def __init__(self, concat_oscil=False):
        super().__init__()
        self.concat_oscil = concat_oscil

    def forward(self, x):
        if self.concat_oscil:
            return torch.cat([hilbert_envelope(x), x], dim=1)
        return hilbert_envelope(x)
