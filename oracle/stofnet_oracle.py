"""CPU restatement of the StofNet forward path (TEST INFRASTRUCTURE ONLY, see
oracle/__init__.py).  Parity status: pinned by tests/test_oracle_golden.py against
golden vectors captured from the reference (tests/golden/make_golden.py).

Written from the behaviour described in SURVEY.md §3.2/§8a, not from the
reference's source text: a functional graph over a plain ``{name: array}``
parameter dict, with an explicit conv-as-shifted-matmul so the same code runs
in float32 (what the reference computes) or float64 (ground truth used to
show both the reference and the HIP path sit within rounding of the truth).

The arithmetic itself lives in third-party PyTorch (ATen mkldnn conv,
unpinned in the reference's requirements.txt:2); the golden vectors tie this
restatement to torch 2.10.0 CPU results.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class OddSemiGlobalRemainder(RuntimeError):
    """Q1: the reference's SemiGlobalBlock pads (p//2, p//2) with
    p = L - 80*floor(L/80) (models/stofnet.py:111-112); for odd p the add at
    models/stofnet.py:115 raises a torch RuntimeError (size mismatch)."""


def _t(a, dtype):
    if isinstance(a, torch.Tensor):
        if a.requires_grad:                 # oracle/train_oracle.py differentiates through the forward
            return a.to('cpu', dtype)
        return a.detach().to('cpu', dtype)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def conv1d_same(x, w, b, pad):
    """Zero-padded stride-1 dilation-1 cross-correlation, the only conv form the
    network uses (models/stofnet.py:23,24,31,88,94)."""
    return F.conv1d(x, w, b, stride=1, padding=pad)


def conv1d_shifted_matmul(x, w, b, pad):
    """Same op spelled as sum over taps of W[:, :, k] @ shift(x, k - pad).
    Independent of ATen's conv kernels; used to cross-check `conv1d_same`
    and as the float64 ground truth."""
    n, ci, L = x.shape
    co, _, k = w.shape
    xp = F.pad(x, (pad, pad))
    out = torch.zeros((n, co, L + 2 * pad - k + 1), dtype=x.dtype)
    for tap in range(k):
        out += torch.einsum('oc,nct->not', w[:, :, tap], xp[:, :, tap:tap + out.shape[-1]])
    return out + b[None, :, None]


def sample_shuffle(x: torch.Tensor, r: int) -> torch.Tensor:
    """utils/sample_shuffle.py:10-28:  out[n, c, w*r + k] = in[n, k*C + c, w],
    C = C_in // r  (channel grouping is (r, C), not nn.PixelShuffle's (C, r))."""
    n, cin, w = x.shape
    c = cin // r
    if c * r != cin:
        raise RuntimeError(f"shape '[{n}, {r}, {c}, {w}]' is invalid for input of size {x.numel()}")
    out = torch.empty((n, c, w * r), dtype=x.dtype)
    for k in range(r):
        out[:, :, k::r] = x[:, k * c:(k + 1) * c, :]
    return out


def semi_global_block(x, p, prefix, scale, dtype, conv=conv1d_same, taps=None):
    """models/stofnet.py:98-117: x + pad(upsample_s(lrelu(conv5(maxpool_s(lrelu(conv5(x)))))))."""
    L = x.shape[-1]
    npool = L // scale
    rem = L - npool * scale
    wc, we = _t(p[prefix + 'contract_conv.weight'], dtype), _t(p[prefix + 'expand_conv.weight'], dtype)
    z = F.leaky_relu(conv(x, wc, _t(p[prefix + 'contract_conv.bias'], dtype), wc.shape[-1] // 2), 0.01)     # :88 padding = k // 2
    if taps is not None:
        taps['sgb_contract'] = z
    # max-pool, floor mode: the tail remainder is dropped (models/stofnet.py:103)
    z = z[..., :npool * scale].reshape(z.shape[0], z.shape[1], npool, scale).amax(-1)
    if taps is not None:
        taps['sgb_pooled'] = z
    z = F.leaky_relu(conv(z, we, _t(p[prefix + 'expand_conv.bias'], dtype), we.shape[-1] // 2), 0.01)
    if taps is not None:
        taps['sgb_expand'] = z
    if rem % 2:
        raise OddSemiGlobalRemainder(
            f"The size of tensor a ({L}) must match the size of tensor b ({npool * scale + 2 * (rem // 2)}) "
            f"at non-singleton dimension 2")
    up = torch.zeros_like(x)
    # nearest upsample out[j] = in[j // scale], shifted right by rem//2 (Q2)
    up[..., rem // 2: rem // 2 + npool * scale] = z.repeat_interleave(scale, dim=-1)
    return x + up


def stofnet_forward(params: dict, x, upsample_factor: int = 4, semi_global_scale: int = 80,
                    dtype=torch.float32, conv=conv1d_same, taps: dict | None = None, num_blocks: int | None = None):
    """models/stofnet.py:42-67.  `params` uses the reference's state_dict names.
    `taps`, if given, receives per-layer checkpoints (conv1 out, SGB out, each
    residual state, the second-last layer's out, conv_last out).  `num_blocks` (models/stofnet.py:11; default: read from
    the parameter names) and the body kernel size (the weights' last dimension, padding='same') are general, as in the
    reference's constructor; residual layers are models/stofnet.py:39's list."""
    p = params
    x = _t(x, dtype)
    W = lambda n: _t(p[n + '.weight'], dtype)
    B = lambda n: _t(p[n + '.bias'], dtype)
    nb = num_blocks if num_blocks is not None else 1 + max(int(k[4:-7]) for k in p
                                                             if k.startswith('conv') and k.endswith('.weight') and k[4:-7].isdigit())
    if nb < 4:
        raise ValueError('models/stofnet.py:60 reads the loop variable of :52: num_blocks < 4 fails in the reference')
    residual_layers = list(range(3, nb - 1, 2)) + [nb - 1, nb]           # :39
    x = F.relu(conv(x, W('conv1'), B('conv1'), 4))                      # :45
    if taps is not None:
        taps['conv1'] = x
    if semi_global_scale != 1:                                            # :48
        x = semi_global_block(x, p, 'semi_global_block.', semi_global_scale, dtype, conv, taps)
    if taps is not None:
        taps['x0'] = x
    res1 = res = x                                                        # :51
    for i in range(2, nb - 1):                                            # :52-58
        w = W(f'conv{i}')
        y = conv(x, w, B(f'conv{i}'), w.shape[-1] // 2)
        if i in residual_layers:        # 13 blocks: i in {3,5,7,9,11}: residual add, no activation
            x = res + y
            res = x
            if taps is not None:
                taps[f'res{i}'] = x
        else:                           # leaky ReLU 0.01
            x = F.leaky_relu(y, 0.01)
    w = W(f'conv{nb - 1}')
    x = res1 + conv(x, w, B(f'conv{nb - 1}'), w.shape[-1] // 2)           # :61-62
    if taps is not None:
        taps[f'conv{nb - 1}'] = x
    x = conv(x, W('conv_last'), B('conv_last'), 1)                        # :65
    if taps is not None:
        taps['conv_last'] = x
    return sample_shuffle(x, upsample_factor)


def flops_per_waveform(L: int, r: int, semi_global_scale: int = 80) -> float:
    """SURVEY.md §8d / BASELINE.md §3 work model."""
    per_sample = 576 + 11 * 28672 + 192 * r
    total = L * per_sample
    if semi_global_scale != 1:
        total += L * 163840 + (L // semi_global_scale) * 163840
    return 2.0 * total
