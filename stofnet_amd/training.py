"""Training step of StofNet on the gfx950 kernels (reference
main.py:204-248): forward with saved activations, the Gaussian-mask loss, the full backward pass,
AdamW, and the DDP-style gradient all-reduce (one flat 2.58 MB bucket over RCCL).

Everything numerical runs in `stofnet_amd/csrc/train.hip` through the C ABI, exact fp32, layer by
layer on channel-last [N][L][C] activations (the backward pass needs every layer's activation, so
the fused inference sweep does not apply).  PyTorch only owns the device buffers.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .stofnet import StofNet

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
BODY_CONVS = tuple(f'conv{i}' for i in range(2, 13))


def gaussian_kernel(size, sigma=1.0):
    """utils/gaussian.py:4-7."""
    x = np.linspace(-size // 2 + 1, size // 2, size)
    k = np.exp(-np.power(x / sigma, 2) / 2)
    return k / np.sum(k)


class _LazyRepack(dict):
    """name -> kernel-layout image of a weight, packed on first use: which images a step needs depends on the route its
    kernels take (the fused SemiGlobalBlock forward packs its own, the sparse backward and conv_last's data-gradient kernel
    read the raw weights) -- in the default C5 step three of five images were packed and never read.  Keys are known up
    front (`name in images` decides the backward route)."""

    def __init__(self, engine, weights, flip):
        super().__init__()
        self._engine, self._weights, self._flip = engine, weights, flip

    def __contains__(self, k):
        return k in self._weights

    def __missing__(self, k):
        img = self[k] = self._engine._repack(self._weights[k], self._flip)
        return img


class TrainEngine:
    """Layer-by-layer forward with saved activations and the full backward pass of StofNet on the gfx950 training
    kernels (`stof_train_*`), on explicit parameter / gradient dictionaries.  Shared by `StofNetTrainer` (fused loss
    kernels + AdamW kernel on one flat buffer) and by the autograd boundary of `StofNet.forward` in train mode
    (`StofNetFunction`: torch computes the loss and owns the optimizer, as in the reference's main.py:221-248)."""

    def __init__(self, dev, r, sgb, precision='fp32', scale=80, num_blocks=13, body_kernel=7):
        if precision not in ('fp32', 'f16x3'):
            raise ValueError("precision must be 'fp32' or 'f16x3'")
        self.prec = 1 if precision == 'f16x3' else 0       # arithmetic of every 64/512-channel convolution: forward, data gradient, weight gradient
        self.dev, self.r, self.sgb = dev, int(r), bool(sgb)
        # models/stofnet.py:11: any num_blocks >= 4 (the reference's forward reads the loop variable of :52 at :60) and any odd
        # body kernel the layer kernels take; the fused sweeps serve the shipped 13 x k7 geometry, everything else runs
        # layer by layer on the channel-last MFMA kernels
        self.nb, self.kb = int(num_blocks), int(body_kernel)
        if self.nb < 4:
            raise NotImplementedError('StofNet: num_blocks < 4 fails in the reference (models/stofnet.py:60)')
        if self.kb not in (1, 3, 5, 7):
            raise NotImplementedError('StofNet: the gfx950 layer kernels take body kernel sizes 1, 3, 5, 7')
        # SemiGlobalBlock geometry (models/stofnet.py:83-85): pool / upsample by `scale`, feat_scale = max(1, scale // 10)
        self.scale = int(scale)
        self.cmid = 64 * max(1, self.scale // 10)
        # split-fp16 mode: conv2..conv12 + conv_last of the forward run as ONE fused sweep that also writes every layer's
        # output for the backward pass (stof_train_sweep) instead of twelve layer launches; STOF_TRAIN_SWEEP=0 keeps the layers
        import os
        self.sweep = (self.prec == 1 and (not self.sgb or self.scale == 80) and self.nb == 13 and self.kb == 7
                      and os.environ.get('STOF_TRAIN_SWEEP', '1') != '0'
                      and os.environ.get('STOF_BODY16', '1') != '0')
        self._sweep_blob = None
        # SemiGlobalBlock backward from the pool's sparse gradient (STOF_TRAIN_SGB_SPARSE=0: dense route, for A/B runs and tests)
        self.sparse_sgb = os.environ.get('STOF_TRAIN_SGB_SPARSE', '1') != '0'
        self.sparse_sgb_dgrad = os.environ.get('STOF_TRAIN_SGB_SPARSE_DGRAD', '1') != '0'
        if self.sgb and not 2 <= self.scale <= 256:
            raise NotImplementedError('SemiGlobalBlock sample_scale must be in [2, 256] for the gfx950 kernels')
        self._gscale = 1.0
        self.g = {}

    # ---- thin wrappers over the C ABI ---------------------------------------------------------
    def _st(self):
        return _lib.stream_ptr(self.dev)

    def _conv(self, x, w_tm, bias, cin, cout, K, act=ACT_NONE, residual=None, saved=None):
        n, L = x.shape[0], x.shape[1]
        y = torch.empty((n, L, cout), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().stof_train_conv(_lib.ptr(x), _lib.ptr(w_tm), _lib.ptr(bias), _lib.ptr(residual),
                                              _lib.ptr(saved), _lib.ptr(y), n, L, cin, cout, K, act, self.prec, self._st()),
                   'stof_train_conv')
        return y

    def _repack(self, w, flip):
        cout, cin, K = w.shape
        lib = _lib.lib()
        out = torch.empty(lib.stof_train_repack_floats(cout, cin, K, 1 if flip else 0, self.prec), dtype=torch.float32, device=self.dev)
        _lib.check(lib.stof_train_repack(_lib.ptr(w), _lib.ptr(out), cout, cin, K, 1 if flip else 0, self.prec, self._st()),
                   'stof_train_repack')
        return out

    def _wgrad(self, x, dy, name, cin, cout, K):
        n, L = x.shape[0], x.shape[1]
        need = _lib.lib().stof_train_wgrad_workspace_bytes(cin, cout, K)
        ws = getattr(self, '_wgrad_ws', None)
        if ws is None or ws.numel() < need:
            ws = self._wgrad_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        _lib.check(_lib.lib().stof_train_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(self.g[name + '.weight']),
                                               _lib.ptr(self.g[name + '.bias']), n, L, cin, cout, K, 1.0 / self._gscale, self.prec, _lib.ptr(ws),
                                               ws.numel(),
                                               self._st()), 'stof_train_wgrad')

    def _add(self, a, b):
        out = torch.empty_like(a)
        _lib.check(_lib.lib().stof_train_add(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.numel(), self._st()), 'stof_train_add')
        return out

    # ---- forward (activations kept) and backward, on explicit parameter / gradient dictionaries --------------
    def _forward_saved(self, p, frame, keep=True):
        """models/stofnet.py:42-67 layer by layer, every activation kept for the backward pass.
        Returns (pred [N, L*r] = conv_last's channel-last output = the sample-shuffled prediction, saved)."""
        lib = _lib.lib()
        _lib.require_device(frame, 'frame')
        r = self.r
        x = frame.detach().reshape(frame.shape[0], frame.shape[-1]).contiguous().float()
        n, L = x.shape
        S, cm = self.scale, self.cmid
        P = L // S if self.sgb else 0
        rem = L - S * P
        if self.sgb and P == 0:
            raise RuntimeError(_lib.status_string(_lib.STOF_ERR_POOL_EMPTY))     # the reference's max_pool1d error
        if self.sgb and rem % 2:
            raise RuntimeError(f'The size of tensor a ({L}) must match the size of tensor b ({L - 1}) at non-singleton dimension 2')
        sg = 'semi_global_block.'
        st = self._st()
        use_sweep = self.sweep and 'conv2.weight' in p
        head = ('semi_global_block.',) if use_sweep else ('conv', 'semi_global_block.')        # layers that still run one by one
        fwd = _LazyRepack(self, {k[:-7]: w for k, w in p.items()
                                 if k.endswith('.weight') and k != 'conv1.weight' and k.startswith(head)}, False)
        sweep_bwd = use_sweep and os.environ.get('STOF_TRAIN_SWEEP_BWD', '1') != '0'       # conv2..conv12 data gradients: one sweep too
        bwd = _LazyRepack(self, {k[:-7]: w for k, w in p.items() if k.endswith('.weight') and k != 'conv1.weight'
                                 and not (sweep_bwd and k[:-7] in BODY_CONVS)}, True) if keep else None
        a1 = torch.empty((n, L, 64), dtype=torch.float32, device=self.dev)
        _lib.check(lib.stof_train_conv1(_lib.ptr(x), _lib.ptr(p['conv1.weight']), _lib.ptr(p['conv1.bias']), _lib.ptr(a1),
                                        n, L, st), 'stof_train_conv1')
        c = pooled = arg = e = None
        if use_sweep:
            # SemiGlobalBlock up to the expand conv on the layer kernels (the backward pass needs c / pooled / arg); its
            # up-sampled map is added inside the sweep, which recomputes relu(conv1) from x
            if self.sgb and os.environ.get('STOF_TRAIN_SGB_FUSED', '1') != '0':
                # contract conv + lrelu + max-pool fused as in inference (no [N, L, 512] tensor), with the pool's arg-max
                pooled = torch.empty((n, P, cm), dtype=torch.float32, device=self.dev)
                arg = torch.empty((n, P, cm), dtype=torch.uint8, device=self.dev)
                if getattr(self, '_sgb_blob', None) is None:
                    self._sgb_blob = torch.empty(lib.stof_train_sgb_blob_bytes(), dtype=torch.uint8, device=self.dev)
                _lib.check(lib.stof_train_sgb_contract_pool(_lib.ptr(p['conv1.weight'].contiguous()), _lib.ptr(p['conv1.bias'].contiguous()),
                                                            _lib.ptr(p[sg + 'contract_conv.weight'].contiguous()),
                                                            _lib.ptr(p[sg + 'contract_conv.bias'].contiguous()), _lib.ptr(self._sgb_blob),
                                                            _lib.ptr(x), _lib.ptr(pooled), _lib.ptr(arg), n, L, st),
                           'stof_train_sgb_contract_pool')
                e = self._conv(pooled, fwd[sg + 'expand_conv'], p[sg + 'expand_conv.bias'], cm, 64, 5, ACT_LRELU)
            elif self.sgb:
                c, pooled, arg, e = self._sgb_head(a1, fwd[sg + 'contract_conv'], p[sg + 'contract_conv.bias'],
                                                   fwd[sg + 'expand_conv'], p[sg + 'expand_conv.bias'])
            desc = _lib.NetDesc(int(r), 80 if self.sgb else 1, _lib.PREC_F16X3, 0)
            import ctypes
            nbytes = lib.stof_train_sweep_blob_bytes(ctypes.byref(desc))
            if self._sweep_blob is None or self._sweep_blob.numel() < nbytes:
                self._sweep_blob = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
            names = ['conv1'] + [f'conv{i}' for i in range(2, 13)] + ['conv_last']
            arr = (ctypes.c_void_p * 26)()
            for i, nm in enumerate(names):
                arr[2 * i] = _lib.ptr(p[nm + '.weight'].contiguous())
                arr[2 * i + 1] = _lib.ptr(p[nm + '.bias'].contiguous())
            _lib.check(lib.stof_train_sweep_pack(ctypes.byref(desc), arr, _lib.ptr(self._sweep_blob), st), 'stof_train_sweep_pack')
            dump = torch.empty(lib.stof_train_sweep_dump_floats(n, L), dtype=torch.float32, device=self.dev)
            z = torch.empty((n, L * r), dtype=torch.float32, device=self.dev)
            # r4: when the backward pass runs as one sweep + one batched weight-gradient launch, the dumps are SPLIT ROWS
            # ([64 x fp16 hi | 64 x fp16 lo] per 256-byte row: the halves the sweeps compute anyway) and the weight-gradient
            # kernel stages them without converting.  STOF_TRAIN_SPLIT_DUMPS=0: fp32 dumps everywhere (A/B runs).
            split = (keep and sweep_bwd and self.prec == 1 and os.environ.get('STOF_TRAIN_WGRAD_BATCH', '1') != '0'
                     and os.environ.get('STOF_TRAIN_SPLIT_DUMPS', '1') != '0')
            if split:
                code = lib.stof_train_sweep_split(ctypes.byref(desc), _lib.ptr(self._sweep_blob), _lib.ptr(x), _lib.ptr(e), _lib.ptr(dump),
                                                  _lib.ptr(z), n, L, st)
                if code == _lib.STOF_ERR_UNSUPPORTED:
                    split = False
                else:
                    _lib.check(code, 'stof_train_sweep_split')
            if not split:
                _lib.check(lib.stof_train_sweep(ctypes.byref(desc), _lib.ptr(self._sweep_blob), _lib.ptr(x), _lib.ptr(e), _lib.ptr(dump),
                                                _lib.ptr(z), n, L, st), 'stof_train_sweep')
            if not keep:
                return z, None
            t = dump[:12 * n * L * 64].view(12, n, L, 64)
            xs = [t[0]] + [t[2 + 2 * k] for k in range(5)]
            ys = [t[1 + 2 * k] for k in range(5)]
            v = {1: xs[0]}
            for k in range(5):
                v[2 * k + 2], v[2 * k + 3] = ys[k], xs[k + 1]
            saved = dict(x=x, a1=a1, c=c, pooled=pooled, arg=arg, e=e, xs=xs, ys=ys, v=v, x6=t[11], bwd=bwd, n=n, L=L, P=P, rem=rem,
                         _dump=dump, split=split, desc=desc, wdev=[p[f'conv{i}.weight'] for i in range(2, 13)],
                         wc=p.get(sg + 'contract_conv.weight'), wl=p['conv_last.weight'], w1=p['conv1.weight'])
            return z, saved
        if self.sgb:
            x0, c, pooled, arg, e = self._sgb_forward(a1, fwd[sg + 'contract_conv'], p[sg + 'contract_conv.bias'],
                                                      fwd[sg + 'expand_conv'], p[sg + 'expand_conv.bias'])
        else:
            x0 = a1
        if not keep:
            del c, pooled, arg, e
            c = pooled = arg = e = None
        # models/stofnet.py:51-62 for any num_blocks: even layers leaky ReLU, odd layers (>= 3) add the running residual
        # and become it; the second-last layer adds res1 = x0.  v[i] = output of conv{i} (v[1] = x0).
        nb, kb = self.nb, self.kb
        v = {1: x0}
        for i in range(2, nb - 1):
            nm = f'conv{i}'
            if i % 2:
                v[i] = self._conv(v[i - 1], fwd[nm], p[nm + '.bias'], 64, 64, kb, ACT_NONE, residual=v[i - 2])
                if not keep:                   # v[i] is the running residual now; x0 stays for the long skip
                    v[i - 1] = None
                    if i > 3:
                        v[i - 2] = None
            else:
                v[i] = self._conv(v[i - 1], fwd[nm], p[nm + '.bias'], 64, 64, kb, ACT_LRELU)
        nm = f'conv{nb - 1}'
        x6 = self._conv(v[nb - 2], fwd[nm], p[nm + '.bias'], 64, 64, kb, ACT_NONE, residual=x0)
        z = self._conv(x6, fwd['conv_last'], p['conv_last.bias'], 64, r, 3, ACT_NONE)      # [N, L, r] == shuffled [N, L*r]
        if not keep:
            return z.view(n, L * r), None
        saved = dict(x=x, a1=a1, c=c, pooled=pooled, arg=arg, e=e, v=v, x6=x6, bwd=bwd, n=n, L=L, P=P, rem=rem,
                     wc=p.get('semi_global_block.contract_conv.weight'), wl=p['conv_last.weight'], w1=p['conv1.weight'])
        return z.view(n, L * r), saved

    def _sgb_head(self, a1, w_contract, b_contract, w_expand, b_expand, width=64, K=5):
        """contract conv -> lrelu -> max-pool -> expand conv -> lrelu (models/stofnet.py:100-107) on channel-last a1
        [N, L, width]; inside StofNet width = 64 and K = 5, the standalone block (models/stofnet.py:80) takes any."""
        lib, st = _lib.lib(), self._st()
        n, L = a1.shape[0], a1.shape[1]
        S = self.scale
        cm = self.cmid if width == 64 else width * max(1, S // 10)
        P = L // S
        c = self._conv(a1, w_contract, b_contract, width, cm, K, ACT_LRELU)
        pooled = torch.empty((n, max(P, 1), cm), dtype=torch.float32, device=self.dev)[:, :P]
        arg = torch.empty((n, max(P, 1), cm), dtype=torch.uint8, device=self.dev)[:, :P]
        _lib.check(lib.stof_train_pool(_lib.ptr(c), _lib.ptr(pooled), _lib.ptr(arg), n, L, P, cm, S, st), 'stof_train_pool')
        e = self._conv(pooled, w_expand, b_expand, cm, width, K, ACT_LRELU)
        return c, pooled, arg, e

    def _sgb_forward(self, a1, w_contract, b_contract, w_expand, b_expand, width=64, K=5):
        """SemiGlobalBlock.forward (models/stofnet.py:98-117) on channel-last a1 [N, L, 64]:
        a1 + pad(upsample(lrelu(expand(maxpool(lrelu(contract(a1))))))).  Returns (out, c, pooled, arg, e)."""
        lib, st = _lib.lib(), self._st()
        n, L = a1.shape[0], a1.shape[1]
        S = self.scale
        P = L // S
        rem = L - S * P
        c, pooled, arg, e = self._sgb_head(a1, w_contract, b_contract, w_expand, b_expand, width, K)
        out = torch.empty_like(a1)
        if width == 64:
            _lib.check(lib.stof_train_upsample_add(_lib.ptr(a1), _lib.ptr(e), _lib.ptr(out), n, L, P, rem // 2, S, st),
                       'stof_train_upsample_add')
        else:
            _lib.check(lib.stof_train_upsample_add_c(_lib.ptr(a1), _lib.ptr(e), _lib.ptr(out), n, L, P, rem // 2, S, width, st),
                       'stof_train_upsample_add_c')
        return out, c, pooled, arg, e

    def _backward_saved(self, saved, dpred, g, gscale, dx=None):
        """Backward pass from dpred [N, L*r] = gscale * dloss/dpred (gscale a power of two: the f16x3 data-gradient
        convolutions would otherwise work on fp16 subnormals; the weight-gradient kernels multiply by 1/gscale, exact).
        Writes every parameter gradient into the tensors of `g` (name -> tensor of the parameter's shape) and, if `dx`
        [N, L] is given, the gradient with respect to the input frame into it."""
        lib = _lib.lib()
        r, st = self.r, self._st()
        sg = 'semi_global_block.'
        n, L, P, rem = saved['n'], saved['L'], saved['P'], saved['rem']
        xs, ys, v, x6, bwd, a1 = saved.get('xs'), saved.get('ys'), saved['v'], saved['x6'], saved['bwd'], saved['a1']
        nb, kb = self.nb, self.kb
        second_last = f'conv{nb - 1}'
        self._gscale = float(gscale)
        self.g = g
        dz = dpred.view(n, L, r)
        self._wgrad(x6, dz, 'conv_last', 64, r, 3)
        # conv_last's data gradient: r input channels would be padded to a 64-channel block by the layer kernels
        g6 = torch.empty((n, L, 64), dtype=torch.float32, device=self.dev)
        split = bool(saved.get('split'))          # the forward dumps are split rows: only the sweep + batched route reads them
        # split route: g6 is a split-row tensor as well (the backward sweep's input, conv12's output gradient for the weight-gradient
        # launch and the long-skip join all take it as such)
        code = (lib.stof_train_conv_last_dgrad_split if split else lib.stof_train_conv_last_dgrad)(
            _lib.ptr(dz.contiguous()), _lib.ptr(saved['wl'].contiguous()), _lib.ptr(g6), n, L, r, st)
        if code == _lib.STOF_ERR_UNSUPPORTED:
            g6 = self._conv(dz, bwd['conv_last'], None, r, 64, 3)
            if split:
                g6f, g6 = g6, torch.empty_like(g6)
                _lib.check(lib.stof_train_to_split_rows(_lib.ptr(g6f), _lib.ptr(g6), n * L, st), 'stof_train_to_split_rows')
        else:
            _lib.check(code, 'stof_train_conv_last_dgrad')
        batch_wgrad = (self.prec == 1 and saved.get('_dump') is not None and second_last not in bwd
                       and (split or os.environ.get('STOF_TRAIN_WGRAD_BATCH', '1') != '0'))
        if split and not batch_wgrad:
            raise RuntimeError('split-row dumps without the backward sweep')
        if not batch_wgrad:
            self._wgrad(v[nb - 2], g6, second_last, 64, 64, kb)
        if saved.get('_dump') is not None and second_last not in bwd:
            # the eleven data-gradient convolutions conv12^T .. conv2^T as ONE backward sweep (stof_train_sweep_bwd), then the
            # weight gradients from its dumps: tensor j odd = dL/dx_k, k = (11 - j) / 2; j even = dL/d(pre-activation of conv(12 - j))
            import ctypes
            nbytes = lib.stof_train_sweep_blob_bytes(ctypes.byref(saved['desc']))
            if getattr(self, '_sweep_blob_bwd', None) is None or self._sweep_blob_bwd.numel() < nbytes:
                self._sweep_blob_bwd = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
            arr = (ctypes.c_void_p * 11)(*[_lib.ptr(w.contiguous()) for w in saved['wdev']])
            _lib.check(lib.stof_train_sweep_bwd_pack(arr, _lib.ptr(self._sweep_blob_bwd), st), 'stof_train_sweep_bwd_pack')
            dumpb = torch.empty(lib.stof_train_sweep_dump_floats(n, L), dtype=torch.float32, device=self.dev)
            _lib.check((lib.stof_train_sweep_bwd_split if split else lib.stof_train_sweep_bwd)(
                ctypes.byref(saved['desc']), _lib.ptr(self._sweep_blob_bwd), _lib.ptr(g6), _lib.ptr(saved['_dump']),
                _lib.ptr(dumpb), n, L, st), 'stof_train_sweep_bwd')
            T = dumpb[:12 * n * L * 64].view(12, n, L, 64)
            pairs = [(xs[5], g6, 'conv12')] if batch_wgrad else []                      # (input activation, output gradient, layer)
            for k in range(4, -1, -1):
                pairs.append((ys[k], T[9 - 2 * k], f'conv{2 * k + 3}'))                 # g_{k+1} = T[11 - 2 (k + 1)]
                pairs.append((xs[k], T[10 - 2 * k], f'conv{2 * k + 2}'))                # u_k
            if batch_wgrad:
                # r4: the eleven k7 weight gradients in ONE launch pair (stof_train_wgrad_batch) instead of eleven launches +
                # eleven reductions of 58 MB of partials each
                import ctypes
                cnt = len(pairs)
                need = lib.stof_train_wgrad_batch_workspace_bytes(cnt, 7)
                ws = getattr(self, '_wgrad_batch_ws', None)
                if ws is None or ws.numel() < need:
                    ws = self._wgrad_batch_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
                arr = lambda ts: (ctypes.c_void_p * cnt)(*[_lib.ptr(t) for t in ts])
                if split:
                    # every x operand is a forward dump tensor (0..10), every dy a backward dump tensor or g6: all split rows
                    all_bits = (1 << cnt) - 1
                    _lib.check(lib.stof_train_wgrad_batch_split(arr([a for a, _, _ in pairs]), arr([d for _, d, _ in pairs]),
                                                                arr([self.g[nm + '.weight'] for _, _, nm in pairs]),
                                                                arr([self.g[nm + '.bias'] for _, _, nm in pairs]), cnt, all_bits,
                                                                all_bits, n, L, 7, 1.0 / self._gscale, _lib.ptr(ws), ws.numel(), st),
                               'stof_train_wgrad_batch_split')
                else:
                    _lib.check(lib.stof_train_wgrad_batch(arr([a for a, _, _ in pairs]), arr([d for _, d, _ in pairs]),
                                                          arr([self.g[nm + '.weight'] for _, _, nm in pairs]),
                                                          arr([self.g[nm + '.bias'] for _, _, nm in pairs]), cnt, n, L, 7,
                                                          1.0 / self._gscale, _lib.ptr(ws), ws.numel(), st), 'stof_train_wgrad_batch')
            else:
                for a, d, nm in pairs:
                    self._wgrad(a, d, nm, 64, 64, 7)
            gg = T[11]                                                                   # dL/dx_0 without the long skip
        else:
            # any num_blocks, layer by layer.  Walking down from the second-last layer: an odd layer i holds the TOTAL gradient
            # gg of its output (its own consumer + the residual add two layers on); its transposed convolution, masked with
            # lrelu'(v[i-1]), is u = the gradient before the activation of the even layer i-1; that layer's transposed
            # convolution plus gg (the residual path) is the total gradient of v[i-2].
            m = nb - 2                                                                # last layer of the loop (:52)
            if m % 2:
                gg, u = self._conv(g6, bwd[second_last], None, 64, 64, kb), None        # d/dv[m], v[m] a residual state
            else:
                gg, u = None, self._conv(g6, bwd[second_last], None, 64, 64, kb, ACT_LRELU, saved=v[m])
            for i in range(m, 1, -1):
                nm = f'conv{i}'
                if i % 2:
                    self._wgrad(v[i - 1], gg, nm, 64, 64, kb)
                    u = self._conv(gg, bwd[nm], None, 64, 64, kb, ACT_LRELU, saved=v[i - 1])
                else:
                    self._wgrad(v[i - 1], u, nm, 64, 64, kb)
                    gg = self._conv(u, bwd[nm], None, 64, 64, kb, residual=gg)      # (gg None for the loop's last layer)
        if split:                                                                 # long skip res1 (models/stofnet.py:62)
            g_x0 = torch.empty((n, L, 64), dtype=torch.float32, device=self.dev)
            _lib.check(lib.stof_train_add_split2(_lib.ptr(gg), _lib.ptr(g6), _lib.ptr(g_x0), n * L, st), 'stof_train_add_split2')
        else:
            g_x0 = self._add(gg, g6)
        if self.sgb and P:
            e, pooled, arg, c = saved['e'], saved['pooled'], saved['arg'], saved['c']
            ge = torch.empty((n, P, 64), dtype=torch.float32, device=self.dev)
            S, cm = self.scale, self.cmid
            _lib.check(lib.stof_train_upsample_bwd(_lib.ptr(g_x0), _lib.ptr(e), _lib.ptr(ge), n, L, P, rem // 2, S, st),
                       'stof_train_upsample_bwd')
            self._wgrad(pooled, ge, sg + 'expand_conv', cm, 64, 5)
            gpool = self._conv(ge, bwd[sg + 'expand_conv'], None, 64, cm, 5)
            # The gradient behind the max-pool is zero except at ONE row per (waveform, window, channel).  Where the kernels
            # take the shape, contract_conv's weight gradient and data gradient come straight from those non-zeros
            # (stof_train_sgb_contract_wgrad / _dgrad); otherwise -- and with STOF_TRAIN_SGB_SPARSE=0 -- the dense [N, L, cm]
            # gradient is built (stof_train_pool_bwd) and goes through the layer kernels like every other convolution.
            U = _lib.STOF_ERR_UNSUPPORTED
            wcode = dcode = U
            g_a1 = None
            if self.sparse_sgb and pooled is not None and saved.get('wc') is not None:
                def scratch(name, need):
                    ws = getattr(self, name, None)
                    if ws is None or ws.numel() < need:
                        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.dev)
                        setattr(self, name, ws)
                    return ws
                ws = scratch('_sgb_wgrad_ws', lib.stof_train_sgb_wgrad_workspace_bytes(cm))
                wcode = lib.stof_train_sgb_contract_wgrad(_lib.ptr(gpool), _lib.ptr(arg), _lib.ptr(pooled), _lib.ptr(a1),
                                                          _lib.ptr(g[sg + 'contract_conv.weight']), _lib.ptr(g[sg + 'contract_conv.bias']),
                                                          n, L, P, cm, S, 1.0 / self._gscale, _lib.ptr(ws), ws.numel(), st)
                if wcode != U:
                    _lib.check(wcode, 'stof_train_sgb_contract_wgrad')
                if self.sparse_sgb_dgrad:
                    ws = scratch('_sgb_dgrad_ws', lib.stof_train_sgb_dgrad_workspace_bytes(cm))
                    g_a1 = torch.empty((n, L, 64), dtype=torch.float32, device=self.dev)
                    dcode = lib.stof_train_sgb_contract_dgrad(_lib.ptr(gpool), _lib.ptr(arg), _lib.ptr(pooled), _lib.ptr(saved['wc'].contiguous()),
                                                              _lib.ptr(g_x0), _lib.ptr(g_a1), n, L, P, cm, S, _lib.ptr(ws), ws.numel(), st)
                    if dcode != U:
                        _lib.check(dcode, 'stof_train_sgb_contract_dgrad')
            self.sgb_sparse_taken = (wcode != U and dcode != U)        # bench.py: which route this backward really ran
            if wcode == U or dcode == U:
                gc = torch.empty((n, L, cm), dtype=torch.float32, device=self.dev)
                _lib.check(lib.stof_train_pool_bwd(_lib.ptr(gpool), _lib.ptr(arg), _lib.ptr(c), _lib.ptr(pooled), _lib.ptr(gc), n, L, P, cm, S, st),
                           'stof_train_pool_bwd')
                if wcode == U:
                    self._wgrad(a1, gc, sg + 'contract_conv', 64, cm, 5)
                if dcode == U:
                    g_a1 = self._conv(gc, bwd[sg + 'contract_conv'], None, cm, 64, 5, residual=g_x0)
        else:
            g_a1 = g_x0
        ws1 = torch.empty(lib.stof_train_conv1_wgrad_workspace_bytes(), dtype=torch.uint8, device=self.dev)
        _lib.check(lib.stof_train_conv1_wgrad(_lib.ptr(saved['x']), _lib.ptr(g_a1), _lib.ptr(a1), _lib.ptr(g['conv1.weight']),
                                              _lib.ptr(g['conv1.bias']), n, L, 1.0 / self._gscale, _lib.ptr(ws1),
                                              ws1.numel(), st), 'stof_train_conv1_wgrad')
        if dx is not None:
            # d loss / d frame (the reference's autograd yields it for free, models/stofnet.py:45): conv1 transposed on the masked gradient
            _lib.check(lib.stof_train_conv1_dgrad(_lib.ptr(g_a1), _lib.ptr(a1), _lib.ptr(saved['w1'].contiguous()), _lib.ptr(dx),
                                                  n, L, 1.0 / self._gscale, st), 'stof_train_conv1_dgrad')


class StofNetFunction(torch.autograd.Function):
    """Autograd boundary of the train-mode forward: `masks_pred = model(frame)` (main.py:221) returns a tensor whose
    `backward()` runs the `stof_train_*` data- and weight-gradient kernels and hands torch the gradient of every
    `nn.Parameter`, so the reference's own lines -- torch loss (main.py:228-232), `optimizer.zero_grad();
    loss.backward(); optimizer.step()` with `optim.AdamW` and `CosineAnnealingLR` (main.py:179-180,246-248,288) -- run
    unchanged.  The gradient with respect to the input frame is provided too when `frame.requires_grad`."""

    @staticmethod
    def forward(ctx, frame, engine, names, *params):
        p = dict(zip(names, params))
        with torch.cuda.device(engine.dev):
            pred, saved = engine._forward_saved({k: v.detach() for k, v in p.items()}, frame)
        ctx.engine, ctx.names, ctx.saved = engine, names, saved
        ctx.shapes = [tuple(v.shape) for v in params]
        n, m = pred.shape
        return pred.view(n, 1, m)

    @staticmethod
    def backward(ctx, grad_out):
        engine, saved = ctx.engine, ctx.saved
        if saved is None:
            raise RuntimeError('Trying to backward through the graph a second time: the saved activations of '
                               'StofNet.forward have been freed')
        n, m = saved['n'], saved['L'] * engine.r
        with torch.cuda.device(engine.dev):
            dpred = grad_out.detach().reshape(n, m).contiguous().float()
            gscale = 1.0
            if engine.prec == 1:
                # dloss/dpred of a mean-reduced loss is ~1e-6: fp16-subnormal for the split-fp16 data-gradient
                # convolutions.  Scale by the power of two that brings the largest entry to [1, 2) (exact); the
                # weight-gradient kernels multiply by 1/scale (exact).  One host read per step.
                # The same read carries the range-guard word of the PREVIOUS backward (below).
                prev = getattr(engine, '_bwd_overflow', None)
                word = dpred.abs().amax().reshape(1)
                if prev is not None:
                    word = torch.cat([word, prev])
                    engine._bwd_overflow = None
                host = word.tolist()
                amax = host[0]
                if prev is not None and host[1] != 0.0:
                    ctx.saved = None
                    raise FloatingPointError("StofNet(train_precision='f16x3'): the previous backward produced a non-finite gradient "
                                             "(a back-propagated value left the fp16 range; its gradients were zeroed); train with "
                                             "train_precision='fp32'")
                if not math.isfinite(amax):
                    ctx.saved = None
                    raise FloatingPointError("StofNet(train_precision='f16x3'): non-finite dloss/dpred (an activation left the fp16 "
                                             "range of the split-fp16 arithmetic?); train with train_precision='fp32'")
                if amax > 0.0 and math.isfinite(amax):
                    gscale = 2.0 ** (-math.floor(math.log2(amax)))
                    dpred = dpred * gscale
            sizes = [int(np.prod(sh)) if len(sh) else 1 for sh in ctx.shapes]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=engine.dev)    # fresh: torch may keep these views as .grad
            g, off = {}, 0
            for name, sh, k in zip(ctx.names, ctx.shapes, sizes):
                g[name] = flat[off:off + k].view(sh)
                off += k
            dx = None
            if ctx.needs_input_grad[0]:                 # d loss / d frame: the reference's autograd yields it (models/stofnet.py:45)
                dx = torch.empty((n, saved['L']), dtype=torch.float32, device=engine.dev)
            engine._backward_saved(saved, dpred, g, gscale, dx)
            if engine.prec == 1:
                # Range guard of the split-fp16 backward, without a second host read: a non-finite gradient zeroes this step's
                # gradients on the device (the optimizer step moves nothing but weight decay) and sets a device word that the NEXT
                # backward reads together with its amax (or StofNet.raise_if_overflow() at any time) -> FloatingPointError.
                bad = ~torch.isfinite(flat).all()
                flat.masked_fill_(bad, 0.0)
                if dx is not None:
                    dx.masked_fill_(bad, 0.0)
                engine._bwd_overflow = bad.float().reshape(1)
        ctx.saved = None
        return (None if dx is None else dx.view(n, 1, saved['L']), None, None) + tuple(g[name] for name in ctx.names)


class StofNetTrainer(TrainEngine):
    """AdamW(lr, weight_decay) + MSE(pred, 20*blur7(onehot(gt))/max) + lambda*mean|pred| on `model`
    (main.py:179,184-188,228-232,246-248).  The model's parameters are re-seated as views of one flat
    buffer so the optimizer kernel and the gradient all-reduce touch a single tensor."""

    def __init__(self, model: StofNet, lr=5e-4, weight_decay=1e-8, lambda_value=1e-2, mask_amplitude=20,
                 kernel_size=7, sigma=1, betas=(0.9, 0.999), eps=1e-8, process_group=None, precision='f16x3'):
        if not model._supported():
            raise NotImplementedError('this StofNet geometry has no gfx950 kernels (see StofNet._supported)')
        if kernel_size != 7:
            raise NotImplementedError('the loss kernel implements the 7-tap blur of config.yaml:23')
        params = list(model.named_parameters())
        super().__init__(params[0][1].device, model.upsample_factor, model.semi_global_block is not None, precision,
                         scale=model.semi_global_scale if model.semi_global_block is not None else 80,
                         num_blocks=model.num_blocks, body_kernel=list(model.kernel_sizes)[1])
        self.model = model
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        self.lam, self.amp = float(lambda_value), float(mask_amplitude)
        self.group = process_group
        self.target_max_hook = None      # tests: stands in for the MAX all-reduce of the blurred-target maximum
        # range guard of the split-fp16 arithmetic: two device ints ([0] number of the last bad step, [1] sticky flag),
        # written by stof_train_adamw_guarded, read by raise_if_overflow()
        self._guard_words = torch.zeros(2, dtype=torch.int32, device=self.dev)
        self._guard_pending = None       # set by _guard_grads: the next step() runs behind the guard (with this loss)
        self.step_count = 0
        dev = self.dev
        _lib.require_device(params[0][1], 'model parameters')
        self.names = [n for n, _ in params]
        sizes = [p.numel() for _, p in params]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.p, self.g = {}, {}
        off = 0
        for (name, prm), n in zip(params, sizes):
            view = self.flat[off:off + n].view(prm.shape)
            view.copy_(prm.data)
            prm.data = view                                   # the module now lives in the flat buffer
            self.p[name] = view
            self.g[name] = self.flat_grad[off:off + n].view(prm.shape)
            off += n
        self._grad_views = self.g
        self.taps = torch.tensor(gaussian_kernel(kernel_size, sigma), dtype=torch.float32, device=dev)

    # ---- forward + loss + backward -----------------------------------------------------------
    def forward_backward(self, frame: torch.Tensor, gt_true: torch.Tensor):
        """frame [N,1,L] fp32, gt_true [N,1,G] int64 (round(gt_sample * r), main.py:218).  Fills the
        gradient buffer and returns (loss as a 0-d float64 device tensor, masks_pred [N,1,L*r])."""
        with torch.cuda.device(self.dev):
            self.flat_grad.zero_()
            pred, saved = self._forward_saved(self.p, frame)
            n, m = pred.shape
            # ---------------- loss (main.py:228-232)
            gt = gt_true.detach().reshape(n, -1).contiguous().to(self.dev, torch.int64)
            target = torch.empty_like(pred)
            dpred = torch.empty_like(pred)
            tmax = torch.empty(1, dtype=torch.float32, device=self.dev)
            loss = torch.empty(1, dtype=torch.float64, device=self.dev)
            # loss scaling by a power of two (exact): dloss/dpred ~ 2*diff/(N*M) would sit in the fp16 subnormal range of
            # the f16x3 data-gradient convolutions; the weight-gradient kernels undo it with 1/scale
            gscale = 2.0 ** math.floor(math.log2(max(n * m / 8.0, 1.0)))
            self._loss_kernels(pred, gt, n, m, gscale, target, tmax, dpred, loss)
            self._backward_saved(saved, dpred, self._grad_views, gscale)
        return loss[0], pred.view(n, 1, m)

    def loss(self, masks_pred: torch.Tensor, gt_true: torch.Tensor) -> torch.Tensor:
        """Loss value only (validation, main.py:322-327) for predictions [N,1,M]."""
        _lib.require_device(masks_pred, 'masks_pred')
        pred = masks_pred.detach().reshape(masks_pred.shape[0], -1).contiguous().float()
        n, m = pred.shape
        gt = gt_true.detach().reshape(n, -1).contiguous().to(self.dev, torch.int64)
        target, dpred = torch.empty_like(pred), torch.empty_like(pred)
        tmax = torch.empty(1, dtype=torch.float32, device=self.dev)
        loss = torch.empty(1, dtype=torch.float64, device=self.dev)
        with torch.cuda.device(self.dev):
            self._loss_kernels(pred, gt, n, m, 1.0, target, tmax, dpred, loss, sharded=False)
        return loss[0]

    def _loss_kernels(self, pred, gt, n, m, gscale, target, tmax, dpred, loss, sharded=True):
        """main.py:228-232.  The blurred target is divided by its maximum over the WHOLE batch (main.py:230): when the
        batch is sharded over ranks the local maxima are MAX-all-reduced between the two kernels (a shard whose echoes
        overlap has a larger maximum than one whose echoes do not)."""
        lib = _lib.lib()
        _lib.check(lib.stof_train_loss_target(_lib.ptr(gt), gt.shape[1], _lib.ptr(self.taps), n, m, _lib.ptr(target),
                                              _lib.ptr(tmax), self._st()), 'stof_train_loss_target')
        if sharded:
            (self.target_max_hook or (lambda t: allreduce_max_(t, self.group)))(tmax)
        _lib.check(lib.stof_train_loss_grad(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(tmax), n, m, self.amp, self.lam,
                                            float(gscale), _lib.ptr(dpred), _lib.ptr(loss), self._st()), 'stof_train_loss_grad')

    def allreduce_grads(self):
        """DDP semantics: average the flat gradient bucket over the process group (RCCL over xGMI on the
        GPU node; one 2.58 MB all-reduce per step, latency-bound)."""
        allreduce_mean_(self.flat_grad, self.group)

    def _guard_grads(self, loss=None):
        """Range guard of the split-fp16 training arithmetic (the reference trains in plain fp32): an activation or a
        back-propagated value beyond the fp16 range turns into inf / NaN and would reach every weight through AdamW.  On the
        device, without a host read: the next `step()` scans the gradient bucket (and `loss`) for non-finite values in front
        of AdamW (stof_train_adamw_guarded: two launches; r4, first form: a dozen torch element-wise launches, ~90 us of a
        4 ms step); if it finds one the gradients count as zero (this step becomes a no-op for the weights apart from weight
        decay) and a sticky flag is set; `raise_if_overflow()` reads it."""
        self._guard_pending = (loss,)

    @property
    def overflow_flag(self):
        return self._guard_words[1:2]

    def raise_if_overflow(self):
        """One host read: raises FloatingPointError if any step since the last check left the fp16 range (its update was
        skipped); use precision='fp32' for such data."""
        if int(self._guard_words[1].item()) != 0:
            self._guard_words[1:2].zero_()
            raise FloatingPointError(f"StofNetTrainer(precision='{'f16x3' if self.prec == 1 else 'fp32'}'): a loss or gradient was "
                                     "non-finite (fp16 range overflow of the split-fp16 arithmetic); those steps were skipped -- "
                                     "train with precision='fp32'")

    def step(self):
        self.step_count += 1
        pending, self._guard_pending = self._guard_pending, None
        with torch.cuda.device(self.dev):
            if pending is not None:
                loss = pending[0]
                if loss is not None and (loss.dtype != torch.float64 or loss.device != self.flat.device):
                    loss = loss.detach().to(self.flat.device, torch.float64)
                _lib.check(_lib.lib().stof_train_adamw_guarded(_lib.ptr(self.flat), _lib.ptr(self.flat_grad), _lib.ptr(self.exp_avg),
                                                               _lib.ptr(self.exp_avg_sq), self.flat.numel(), self.lr, self.betas[0],
                                                               self.betas[1], self.eps, self.wd, self.step_count, _lib.ptr(loss),
                                                               _lib.ptr(self._guard_words), self._st()), 'stof_train_adamw_guarded')
            else:
                _lib.check(_lib.lib().stof_train_adamw(_lib.ptr(self.flat), _lib.ptr(self.flat_grad), _lib.ptr(self.exp_avg),
                                                       _lib.ptr(self.exp_avg_sq), self.flat.numel(), self.lr, self.betas[0],
                                                       self.betas[1], self.eps, self.wd, self.step_count, self._st()),
                           'stof_train_adamw')
        self.model._packed = {}                                # inference weights must be repacked

    def train_step(self, frame, gt_true):
        loss, pred = self.forward_backward(frame, gt_true)
        if _collectives_on(self.group):
            self.allreduce_grads()
        if self.prec == 1:                 # split-fp16: device-side range guard (after the all-reduce: every rank decides alike)
            self._guard_grads(loss)
        self.step()
        return loss, pred

    def train_step_graphed(self, frame, gt_true):
        """`train_step` with forward + loss + backward replayed from ONE hipGraph per
        (frame shape, gt shape): the step is ~65 kernel launches, and at the reference's own batch size (config.yaml:11:
        4 waveforms) the launches, not the kernels, set its duration.  The first call for a shape runs two eager passes
        on a side stream (lazy workspaces, LDS limits and device queries happen there) and captures; later calls copy
        `frame` / `gt_true` into the graph's input buffers and replay.  The range guard + AdamW stay ordinary launches behind the graph
        (its bias correction takes the step count, and `lr` may change per epoch: host scalars).  The returned loss and
        prediction are the graph's output buffers: the next call overwrites them.  With a process group of more than one
        rank (collectives between the kernels) this is `train_step`."""
        if _collectives_on(self.group) or self.target_max_hook is not None:
            return self.train_step(frame, gt_true)
        _lib.require_device(frame, 'frame')
        key = (tuple(frame.shape), tuple(gt_true.shape))
        graphs = self.__dict__.setdefault('_step_graphs', {})
        hit = graphs.get(key)
        with torch.cuda.device(self.dev):
            if hit is None:
                sf = frame.detach().to(self.dev, torch.float32).clone()
                sg = gt_true.detach().to(self.dev, torch.int64).clone()
                side = torch.cuda.Stream(self.dev)
                side.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self.forward_backward(sf, sg)
                torch.cuda.current_stream(self.dev).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    loss, pred = self.forward_backward(sf, sg)
                hit = graphs[key] = (graph, sf, sg, loss, pred)
            graph, sf, sg, loss, pred = hit
            sf.copy_(frame.detach().reshape(sf.shape))
            sg.copy_(gt_true.detach().reshape(sg.shape))
            graph.replay()
        if self.prec == 1:
            self._guard_grads(loss)            # (the guard's scan takes the step number, a host scalar: it runs behind the graph)
        self.step()
        return loss, pred

    def set_lr_cosine(self, epoch, epochs, base_lr):
        """CosineAnnealingLR(optimizer, epochs) stepped once per epoch (main.py:180,288)."""
        self.lr = 0.5 * base_lr * (1.0 + math.cos(math.pi * epoch / epochs))


def _collectives_on(group=None) -> bool:
    """A process group of more than one rank -- or of one rank with STOF_FORCE_COLLECTIVES=1, which lets a one-GPU box
    exercise the RCCL calls themselves (tests/test_rccl_one_rank.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get('STOF_FORCE_COLLECTIVES') == '1'


def allreduce_max_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place MAX all-reduce (the batch-global maximum of the blurred target, main.py:230)."""
    if _collectives_on(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def allreduce_mean_(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean all-reduce of a flat gradient bucket (works on any backend: nccl = RCCL, gloo in tests)."""
    if _collectives_on(group):
        world = dist.get_world_size(group)
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        if world > 1:
            flat_grad.div_(world)
    return flat_grad
