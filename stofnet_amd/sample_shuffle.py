"""SampleShuffle1D on the gfx950 kernel (mirrors utils/sample_shuffle.py:6-28)."""
import torch
import torch.nn as nn

from . import _lib


def sample_shuffle(x: torch.Tensor, upsample_factor: int) -> torch.Tensor:
    _lib.require_device(x, 'x')
    n, cin, w = x.shape
    r = int(upsample_factor)
    if cin % r != 0:
        # the reference's .view raises RuntimeError (utils/sample_shuffle.py:24)
        raise RuntimeError(f"shape '[{n}, {r}, {cin // r}, {w}]' is invalid for input of size {x.numel()}")
    # a pure permutation of elements (utils/sample_shuffle.py:24-27: view / permute / contiguous): every dtype goes through the
    # kernel bit for bit by its element size -- int64 ramps beyond 2^24, float64, float16, bool, complex alike
    xc = x.contiguous()
    out = torch.empty((n, cin // r, w * r), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().stof_sample_shuffle_bytes(_lib.ptr(xc), _lib.ptr(out), n, cin, w, r, x.element_size(),
                                                        _lib.stream_ptr(x.device)), 'stof_sample_shuffle')
    return out


class _ShuffleFn(torch.autograd.Function):
    """Forward on the kernel; the backward of a permutation is its inverse: out[n, c, w r + k] = in[n, k C + c, w]
    => d in[n, k C + c, w] = d out[n, c, w r + k] (a view / permute of the incoming gradient, off the hot path)."""

    @staticmethod
    def forward(ctx, x, r):
        ctx.r = r
        return sample_shuffle(x, r)

    @staticmethod
    def backward(ctx, g):
        r = ctx.r
        n, c, m = g.shape
        w = m // r
        return g.reshape(n, c, w, r).permute(0, 3, 1, 2).reshape(n, r * c, w).contiguous(), None


class SampleShuffle1D(nn.Module):
    def __init__(self, upsample_factor):
        super().__init__()
        self.upsample_factor = upsample_factor

    def forward(self, x):
        if x.requires_grad and torch.is_grad_enabled():
            return _ShuffleFn.apply(x, self.upsample_factor)
        return sample_shuffle(x, self.upsample_factor)
