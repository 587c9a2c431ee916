"""StofNet with the reference's constructor, parameter names and forward signature
(models/stofnet.py:9-117), executing on the gfx950 kernels through the C ABI.

The module owns ordinary nn.Conv1d parameter holders so that the shipped
checkpoints load with `load_state_dict(strict=True)` (main.py:176-177) and
`model.parameters()` / `.to(device)` / `.eval()` behave as in the reference.
`forward` never calls those convs: it repacks the parameters once into the
kernels' streaming layout (re-done when a parameter changes) and calls
`stof_forward*`.

In train mode with gradients enabled (`model.train()`, no `torch.no_grad()`: main.py:205,221) `forward` goes through
an autograd boundary instead (`training.StofNetFunction`): the layer-by-layer training kernels keep their activations
and `loss.backward()` runs the hand-written data-/weight-gradient kernels, delivering `.grad` for every nn.Parameter,
so the reference's torch loss and `optim.AdamW` lines (main.py:228-248) work unchanged.  `train_precision`
('fp32' | 'f16x3') selects the MFMA mode of that path.

`precision` (not in the reference, whose arithmetic is ATen fp32) selects the MFMA mode:
  'auto'  (default) split-fp16 x3 operands with fp32 accumulation -- fp32-level accuracy at 3x the
          fp32 MFMA rate -- plus a device-side range guard: if an activation leaves the fp16 range the
          same call re-runs in exact fp32, decided on the GPU without a host sync (`stof_forward_auto`);
  'fp32'  exact fp32 MFMA, the parity baseline;
  'f16x3' the fast mode alone; `raise_if_overflow()` reports a range overflow.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.init as init

from . import _lib
from .sample_shuffle import SampleShuffle1D

_PRECISIONS = {'fp32': _lib.PREC_FP32, 'f16x3': _lib.PREC_F16X3, 'auto': _lib.PREC_F16X3}


class SemiGlobalBlock(nn.Module):
    """Parameter holder mirroring models/stofnet.py:80-96 (same attribute names)."""

    def __init__(self, in_channels, out_channels, sample_scale=2, kernel_size=5):
        super().__init__()
        self.sample_scale = sample_scale
        self.feat_scale = max(1, sample_scale // 10)
        self.contract_conv = nn.Conv1d(in_channels, self.feat_scale * out_channels, kernel_size=kernel_size,
                                       stride=1, padding=kernel_size // 2)
        self.contract_relu = nn.LeakyReLU()
        self.contract_pool = nn.MaxPool1d(kernel_size=sample_scale, stride=sample_scale)
        self.expand_conv = nn.Conv1d(self.feat_scale * out_channels, out_channels, kernel_size=kernel_size,
                                     stride=1, padding=kernel_size // 2)
        self.expand_relu = nn.LeakyReLU()
        self.expand_upsample = nn.Upsample(scale_factor=sample_scale, mode='nearest')

    def forward(self, x):
        """models/stofnet.py:98-117 standalone: x [N, C, L] -> x + pad(upsample(lrelu(expand(maxpool(lrelu(contract(x))))))).
        Inside StofNet the block is fused into the network kernels; this entry serves direct callers on the same
        channel-last MFMA convolution, pooling and upsample-add kernels the training path uses (exact fp32).  The NCL <->
        channel-last transposes at the boundary are torch copies.  Any C = in_channels = out_channels (the add at :115
        needs them equal), any odd kernel size up to 9, sample_scale 2..256; no autograd graph."""
        from .training import TrainEngine
        _lib.require_device(x, 'x')
        cin = self.contract_conv.in_channels
        if x.dim() != 3 or x.shape[1] != cin:
            raise RuntimeError(f'expected input [N, {cin}, L], got {list(x.shape)}')
        K = int(self.contract_conv.kernel_size[0])
        if self.expand_conv.out_channels != cin or K % 2 == 0 or K > 9:
            raise NotImplementedError('SemiGlobalBlock.forward: the gfx950 kernels serve in_channels == out_channels and odd '
                                      'kernel sizes up to 9')
        n, _, L = x.shape
        S = int(self.sample_scale)
        if L // S == 0:
            raise RuntimeError(_lib.status_string(_lib.STOF_ERR_POOL_EMPTY))
        p = L - L // S * S
        if p % 2:
            raise RuntimeError(f'The size of tensor a ({L}) must match the size of tensor b ({L - 1}) at non-singleton dimension 2')
        eng = TrainEngine(x.device, 1, True, 'fp32', scale=S)
        with torch.cuda.device(x.device):
            a = x.detach().float().permute(0, 2, 1).contiguous()
            out = eng._sgb_forward(a, eng._repack(self.contract_conv.weight.detach(), False), self.contract_conv.bias.detach(),
                                   eng._repack(self.expand_conv.weight.detach(), False), self.expand_conv.bias.detach(),
                                   width=cin, K=K)[0]
            return out.permute(0, 2, 1).contiguous()


class StofNet(nn.Module):

    def __init__(self, upsample_factor=4, num_features=64, num_blocks=13, kernel_sizes=[9, 7, 3], in_channels=1,
                 semi_global_scale=80, weights_init=False, precision='auto', train_precision='f16x3'):
        super().__init__()
        self.num_blocks = num_blocks
        self.in_channels = in_channels
        self.num_features = num_features
        self.kernel_sizes = kernel_sizes
        self.upsample_factor = upsample_factor
        self.semi_global_scale = semi_global_scale
        if precision not in _PRECISIONS:
            raise ValueError(f'precision must be one of {sorted(_PRECISIONS)}')
        self.precision = precision
        if train_precision not in ('fp32', 'f16x3'):
            raise ValueError("train_precision must be 'fp32' or 'f16x3'")
        self.train_precision = train_precision
        self._engines = {}

        self.conv1 = nn.Conv1d(in_channels, num_features, kernel_sizes[0], 1, 4)
        self.conv_last = nn.Conv1d(num_features, upsample_factor, kernel_sizes[-1], 1, 1)
        self.semi_global_block = (SemiGlobalBlock(num_features, num_features, semi_global_scale)
                                  if semi_global_scale != 1 else None)
        for i in range(2, num_blocks):
            setattr(self, f'conv{i}', nn.Conv1d(num_features, num_features, kernel_sizes[1], 1, padding='same'))
        self.sample_shuffle = SampleShuffle1D(upsample_factor)
        self.residual_layers = list(range(3, num_blocks - 1, 2)) + [num_blocks - 1, num_blocks]
        if weights_init:
            self._initialize_weights()
        self._packed = {}
        self._workspace = None
        self._status = None
        self._sticky_status = None        # range-guard word of forward_onsets(sync=False): ORed by the kernels, cleared by the caller

    # ---- kernel-side state -------------------------------------------------
    def _supported(self):
        """models/stofnet.py:11 takes any geometry; the gfx950 kernels take any `num_blocks` >= 4 (fewer fail in the
        reference too: its :60 reads the loop variable of :52) and body kernel sizes 1 / 3 / 5 / 7, with 64 features, one
        input channel, a 9-tap first and a 3-tap last layer (their paddings 4 and 1 are fixed at :23-24)."""
        ks = list(self.kernel_sizes)
        return (self.num_features == 64 and self.num_blocks >= 4 and len(ks) == 3 and ks[0] == 9 and ks[2] == 3
                and ks[1] in (1, 3, 5, 7)
                and self.in_channels == 1 and (self.semi_global_scale == 1 or 2 <= self.semi_global_scale <= 256)
                and 1 <= self.upsample_factor <= 64)

    def _fused_sweep(self):
        """The persistent LDS-resident sweep serves the shipped geometry (13 blocks, 7-tap body, no SemiGlobalBlock or
        sample_scale 80); any other semi_global_scale / num_blocks / body kernel size runs layer by layer on the
        channel-last MFMA kernels of the training path."""
        return self.semi_global_scale in (1, 80) and self.num_blocks == 13 and list(self.kernel_sizes) == [9, 7, 3]

    def _param_list(self):
        ps = [self.conv1.weight, self.conv1.bias]
        for i in range(2, 13):
            c = getattr(self, f'conv{i}')
            ps += [c.weight, c.bias]
        ps += [self.conv_last.weight, self.conv_last.bias]
        if self.semi_global_block is not None:
            ps += [self.semi_global_block.contract_conv.weight, self.semi_global_block.contract_conv.bias,
                   self.semi_global_block.expand_conv.weight, self.semi_global_block.expand_conv.bias]
        return ps

    def _desc(self, prec=None):
        # 4th field: segment policy of the body sweep (0 = automatic; k+1 forces 2^k segments per waveform -- tests)
        return _lib.NetDesc(int(self.upsample_factor), int(self.semi_global_scale),
                            _PRECISIONS[self.precision] if prec is None else prec, int(getattr(self, '_seg_policy', 0)))

    def _packed_weights(self, device, prec=None):
        """Device blob of the parameters in the kernels' layout for MFMA mode `prec`; rebuilt when a parameter changes."""
        prec = _PRECISIONS[self.precision] if prec is None else prec
        ps = self._param_list()
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in ps)
        hit = self._packed.get(prec)
        if hit is None or hit[0] != key:
            lib = _lib.lib()
            desc = self._desc(prec)
            nbytes = lib.stof_packed_weights_bytes(ctypes.byref(desc))
            if nbytes == 0:
                raise NotImplementedError('this StofNet configuration is not supported by the gfx950 kernels')
            host = [np.ascontiguousarray(p.detach().to('cpu', torch.float32).numpy()) for p in ps]
            arr = (ctypes.c_void_p * _lib.NUM_PARAMS)()
            for i, h in enumerate(host):
                arr[i] = h.ctypes.data
            blob = np.empty(nbytes, dtype=np.uint8)
            _lib.check(lib.stof_pack_weights(ctypes.byref(desc), arr, blob.ctypes.data, nbytes), 'stof_pack_weights')
            self._packed[prec] = (key, torch.from_numpy(blob).to(device))
        return self._packed[prec][1]

    # ---- forward -------------------------------------------------------------
    def forward(self, x, _events=None):
        if not self._supported():
            raise NotImplementedError('the gfx950 kernels take 64 features, 1 input channel, kernel_sizes [9, 1|3|5|7, 3], '
                                      'num_blocks >= 4, semi_global_scale 1 or 2..256')
        _lib.require_device(x, 'x')
        if x.dim() != 3 or x.shape[1] != self.in_channels:
            raise RuntimeError(f'expected input [N, {self.in_channels}, L], got {list(x.shape)}')
        n, _, L = x.shape
        r = int(self.upsample_factor)
        if (torch.is_grad_enabled() and _events is None and
                ((self.training and any(p.requires_grad for p in self.parameters())) or x.requires_grad)):
            return self._forward_with_graph(x)
        if not self._fused_sweep():
            return self._forward_layerwise(x)
        xc = x.detach().contiguous().float()
        y = torch.empty((n, 1, L * r), dtype=torch.float32, device=x.device)
        lib = _lib.lib()
        desc = self._desc()
        packed = self._packed_weights(x.device)
        ws_bytes = lib.stof_forward_workspace_bytes(ctypes.byref(desc), n, L)
        if self._workspace is None or self._workspace.numel() < ws_bytes or self._workspace.device != x.device:
            self._workspace = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
        if self.precision != 'fp32' and (self._status is None or self._status.device != x.device):
            self._status = torch.zeros(1, dtype=torch.int32, device=x.device)      # range-guard word of the split-fp16 modes
        with torch.cuda.device(x.device):
            if self.precision == 'auto':
                packed32 = self._packed_weights(x.device, _lib.PREC_FP32)
                code = lib.stof_forward_auto(ctypes.byref(desc), _lib.ptr(packed), _lib.ptr(packed32), _lib.ptr(xc),
                                             _lib.ptr(y), n, L, _lib.ptr(self._workspace), self._workspace.numel(),
                                             _lib.stream_ptr(x.device), _lib.ptr(self._status), _events)
            elif _events is None and self.precision == 'f16x3':
                code = lib.stof_forward_checked(ctypes.byref(desc), _lib.ptr(packed), _lib.ptr(xc), _lib.ptr(y), n, L,
                                                _lib.ptr(self._workspace), self._workspace.numel(),
                                                _lib.stream_ptr(x.device), _lib.ptr(self._status))
            elif _events is None:
                code = lib.stof_forward(ctypes.byref(desc), _lib.ptr(packed), _lib.ptr(xc), _lib.ptr(y), n, L,
                                        _lib.ptr(self._workspace), self._workspace.numel(),
                                        _lib.stream_ptr(x.device))
            else:       # bench.py instrumentation: HIP events around each kernel on the launch stream
                code = lib.stof_forward_events(ctypes.byref(desc), _lib.ptr(packed), _lib.ptr(xc), _lib.ptr(y), n, L,
                                               _lib.ptr(self._workspace), self._workspace.numel(),
                                               _lib.stream_ptr(x.device), _events,
                                               _lib.ptr(self._status) if self.precision == 'f16x3' else None)
        if code == _lib.STOF_ERR_ODD_SGB_REMAINDER:
            # same failure as models/stofnet.py:115
            got = L // 80 * 80 + 2 * ((L - L // 80 * 80) // 2)
            raise RuntimeError(f'The size of tensor a ({L}) must match the size of tensor b ({got}) at '
                               f'non-singleton dimension 2')
        if code == _lib.STOF_ERR_POOL_EMPTY:
            raise RuntimeError(_lib.status_string(code))      # the reference's message (models/stofnet.py:103)
        _lib.check(code, 'stof_forward')
        return y

    def _engine(self, dev, precision):
        from .training import TrainEngine
        key = (str(dev), precision)
        if key not in self._engines:
            self._engines[key] = TrainEngine(dev, self.upsample_factor, self.semi_global_block is not None, precision,
                                             scale=self.semi_global_scale if self.semi_global_block is not None else 80,
                                             num_blocks=self.num_blocks, body_kernel=list(self.kernel_sizes)[1])
        return self._engines[key]

    def _forward_layerwise(self, x):
        """Inference for a semi_global_scale other than 80, a num_blocks other than 13 or a body kernel other than 7
        (models/stofnet.py:11 accepts any): every layer on the
        channel-last MFMA kernels, activations dropped as soon as the next layer has consumed them.  'auto' maps to the
        exact fp32 mode here (the range guard lives in the fused sweep)."""
        if x.shape[0] == 0:
            return torch.empty((0, 1, x.shape[-1] * int(self.upsample_factor)), dtype=torch.float32, device=x.device)
        eng = self._engine(x.device, 'f16x3' if self.precision == 'f16x3' else 'fp32')
        with torch.cuda.device(x.device):
            pred, _ = eng._forward_saved({n: p.detach() for n, p in self.named_parameters()}, x, keep=False)
        return pred.view(x.shape[0], 1, -1)

    def _forward_with_graph(self, x):
        """Train-mode forward (main.py:221): same numbers as the inference path to fp32 rounding, but every layer's
        activation is kept and the result carries a grad_fn whose backward fills the parameters' gradients."""
        from .training import StofNetFunction
        if x.shape[0] == 0:
            raise RuntimeError('StofNet: empty batch in train mode')
        named = list(self.named_parameters())
        dev = named[0][1].device
        if dev != x.device:
            raise RuntimeError(f'Input type ({x.device}) and weight type ({dev}) should be the same')
        return StofNetFunction.apply(x, self._engine(dev, self.train_precision), tuple(n for n, _ in named), *[p for _, p in named])

    # ---- forward with the arg-max picker fused into the sweep ---------------------------------
    def forward_onsets(self, x, window_size=20, return_map=False, cap=32, sync=True):
        """`model(x)` followed by `get_maxima_positions(., window_size, threshold=None)` (main.py:314 -> 320 with th=Null)
        in one pass: returns (counts[N] int32, idx[N, Kmax] int32) -- the integer onset sample indices of every row, ties
        and all, identical to `onset_indices(model(x), window_size)` -- and the map too if `return_map`.  Without the map
        the network output never touches HBM (4 bytes per onset instead of 4*L*r per waveform).  The picker lives in the
        split-fp16 sweep's 16-channel conv_last tile: for precision='fp32' or upsample_factor > 16 (and for an input
        that overflows the fp16 range in 'auto' mode) this falls back to forward() + the picker kernel.

        sync=False (serving loops, bench.py): no host read at all -- returns (counts[N], idx[N, cap]) as launched; entries
        beyond a row's count are undefined and a count above `cap` means the row's list is truncated (the caller's check:
        `(counts > cap).any()`).  The fp16-range guard of these calls is STICKY: the kernels OR into one word that is never
        cleared by a call, so `onsets_overflowed()` -- one host read, whenever the caller likes -- tells whether ANY call
        since the last check left the fp16 range (their onsets are then not to be trusted: re-run those inputs with
        sync=True or precision='fp32'); no exact-fp32 re-run happens in this mode."""
        from .mask2samples import onset_indices
        _lib.require_device(x, 'x')
        if x.dim() != 3 or x.shape[1] != self.in_channels:
            raise RuntimeError(f'expected input [N, {self.in_channels}, L], got {list(x.shape)}')
        n, _, L = x.shape
        r = int(self.upsample_factor)
        lib = _lib.lib()

        def via_map():
            y = self.forward(x)
            counts, idx = onset_indices(y, window_size, None)
            return (counts, idx, y) if return_map else (counts, idx)

        if not self._supported() or not self._fused_sweep() or self.precision == 'fp32' or r > 16 or L < 32 or n == 0:
            return via_map()
        desc = self._desc(_lib.PREC_F16X3)
        xc = x.detach().contiguous().float()
        packed = self._packed_weights(x.device, _lib.PREC_F16X3)
        ws_bytes = lib.stof_forward_onsets_workspace_bytes(ctypes.byref(desc), n, L)
        if self._workspace is None or self._workspace.numel() < ws_bytes or self._workspace.device != x.device:
            self._workspace = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
        if self._status is None or self._status.device != x.device:
            self._status = torch.zeros(1, dtype=torch.int32, device=x.device)
        if not sync and (self._sticky_status is None or self._sticky_status.device != x.device):
            self._sticky_status = torch.zeros(1, dtype=torch.int32, device=x.device)
        y = torch.empty((n, 1, L * r), dtype=torch.float32, device=x.device) if return_map else None
        while True:
            counts = torch.empty((n,), dtype=torch.int32, device=x.device)
            idx = torch.empty((n, cap), dtype=torch.int32, device=x.device)
            if sync:
                self._status.zero_()
            # sync=False: the sticky word, never zeroed here (the kernel ORs into it); see onsets_overflowed()
            status = self._status if sync else self._sticky_status
            with torch.cuda.device(x.device):
                code = lib.stof_forward_onsets(ctypes.byref(desc), _lib.ptr(packed), _lib.ptr(xc), _lib.ptr(y), n, L,
                                               int(window_size), _lib.ptr(counts), _lib.ptr(idx), cap,
                                               _lib.ptr(self._workspace), self._workspace.numel(),
                                               _lib.stream_ptr(x.device), _lib.ptr(status))
            if code == _lib.STOF_ERR_UNSUPPORTED:
                return via_map()
            if code == _lib.STOF_ERR_ODD_SGB_REMAINDER or code == _lib.STOF_ERR_POOL_EMPTY:
                return via_map()                                  # raises the reference's error
            _lib.check(code, 'stof_forward_onsets')
            if not sync:
                return (counts, idx, y) if return_map else (counts, idx)
            kmax = int(torch.maximum(counts.max(), self._status[0] * (cap + 1)))   # the reference's host sync (mask2samples.py:93)
            if int(self._status.item()):                          # an activation left the fp16 range
                if self.precision == 'f16x3':
                    raise FloatingPointError("StofNet(precision='f16x3'): an activation exceeded the fp16 range")
                saved, self.precision = self.precision, 'fp32'
                try:
                    return via_map()
                finally:
                    self.precision = saved
            if kmax <= cap:
                break
            cap = kmax                                            # a row with more ties than the buffer holds: once more
        idx = idx[:, :kmax]
        return (counts, idx, y) if return_map else (counts, idx)

    def onsets_overflowed(self, clear=True) -> bool:
        """forward_onsets(sync=False): synchronise and tell whether ANY such call since the last check produced a non-finite
        value (an activation left the fp16 range); clears the sticky word unless clear=False."""
        if self._sticky_status is None:
            return False
        hit = int(self._sticky_status.item()) != 0
        if hit and clear:
            self._sticky_status.zero_()
        return hit

    def fell_back_to_fp32(self) -> bool:
        """'auto' mode: synchronise and tell whether the LAST forward took the exact-fp32 re-run."""
        return self.precision == 'auto' and self._status is not None and int(self._status.item()) != 0

    def raise_if_overflow(self):
        """f16x3 mode only: synchronise and raise if a forward since the last check produced non-finite
        values (an activation left the fp16 range); re-run such inputs with precision='fp32' (or use 'auto')."""
        if self.precision == 'f16x3' and self._status is not None and int(self._status.item()) != 0:
            self._status.zero_()
            raise FloatingPointError("StofNet(precision='f16x3'): an activation exceeded the fp16 range; "
                                     "use precision='fp32' for this input")
        # train_precision='f16x3' behind loss.backward(): a backward since the last check produced a non-finite gradient (that
        # step's gradients were zeroed on the device).  The next backward makes the same check in the host read it needs anyway.
        for eng in self._engines.values():
            word = getattr(eng, '_bwd_overflow', None)
            if word is not None:
                eng._bwd_overflow = None
                if float(word.item()) != 0.0:
                    raise FloatingPointError("StofNet(train_precision='f16x3'): a backward produced a non-finite gradient (a "
                                             "back-propagated value left the fp16 range; its gradients were zeroed); train with "
                                             "train_precision='fp32'")

    def _initialize_weights(self):
        """models/stofnet.py:69-77."""
        for i in range(1, self.num_blocks):
            if i not in self.residual_layers:
                init.orthogonal_(getattr(self, f'conv{i}').weight, init.calculate_gain('relu'))
            else:
                init.orthogonal_(getattr(self, f'conv{i}').weight)
        init.orthogonal_(self.conv_last.weight)
