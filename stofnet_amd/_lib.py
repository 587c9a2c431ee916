"""ctypes binding of libstofnet_amd.so (the C ABI in include/stofnet_amd.h).

There is no CPU or PyTorch fallback: if the HIP library is missing or a tensor is
not on a ROCm device the call raises.  PyTorch is used only for device memory and
streams.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('STOF_LIB_PATH') or os.path.join(_HERE, 'libstofnet_amd.so')   # override: ablation builds only

STOF_OK = 0
STOF_ERR_BAD_ARG = 1
STOF_ERR_ODD_SGB_REMAINDER = 2
STOF_ERR_UNSUPPORTED = 3
STOF_ERR_WORKSPACE = 4
STOF_ERR_HIP = 5
STOF_ERR_CHANNELS = 6
STOF_ERR_POOL_EMPTY = 7

PREC_FP32 = 0
PREC_F16X3 = 1
NUM_PARAMS = 30


class NetDesc(ctypes.Structure):
    _fields_ = [('upsample_factor', ctypes.c_int32), ('semi_global_scale', ctypes.c_int32),
                ('precision', ctypes.c_int32), ('seg_policy', ctypes.c_int32)]


class StofnetLibraryMissing(ImportError):
    pass


_c = ctypes
_SIGNATURES = {
    'stof_status_string': (_c.c_char_p, [_c.c_int]),
    'stof_abi_version': (_c.c_int, []),
    'stof_packed_weights_bytes': (_c.c_size_t, [_c.POINTER(NetDesc)]),
    'stof_pack_weights': (_c.c_int, [_c.POINTER(NetDesc), _c.POINTER(_c.c_void_p), _c.c_void_p, _c.c_size_t]),
    'stof_forward_workspace_bytes': (_c.c_size_t, [_c.POINTER(NetDesc), _c.c_int64, _c.c_int64]),
    'stof_forward': (_c.c_int, [_c.POINTER(NetDesc), _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_forward_checked': (_c.c_int, [_c.POINTER(NetDesc), _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                        _c.c_int64, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p]),
    'stof_forward_events': (_c.c_int, [_c.POINTER(NetDesc), _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                       _c.c_int64, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.POINTER(_c.c_void_p),
                                       _c.c_void_p]),
    'stof_forward_onsets_workspace_bytes': (_c.c_size_t, [_c.POINTER(NetDesc), _c.c_int64, _c.c_int64]),
    'stof_forward_onsets': (_c.c_int, [_c.POINTER(NetDesc), _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                       _c.c_int32, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                                       _c.c_void_p]),
    'stof_forward_auto': (_c.c_int, [_c.POINTER(NetDesc), _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                     _c.c_int64, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                                     _c.POINTER(_c.c_void_p)]),
    'stof_events_create': (_c.c_int, [_c.c_int32, _c.POINTER(_c.c_void_p)]),
    'stof_events_destroy': (_c.c_int, [_c.c_int32, _c.POINTER(_c.c_void_p)]),
    'stof_event_elapsed_ms': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_float)]),
    'stof_sample_shuffle': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int32,
                                       _c.c_void_p]),
    'stof_sample_shuffle_bytes': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int32,
                                             _c.c_int32, _c.c_void_p]),
    'stof_pick_maxima': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_float,
                                    _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p]),
    'stof_indices_to_coords': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_float,
                                          _c.c_void_p, _c.c_void_p]),
    'stof_reduce_echoes': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                      _c.c_int64, _c.c_float, _c.c_void_p, _c.c_void_p]),
    'stof_hilbert_workspace_bytes': (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    'stof_hilbert': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_hilbert_f64_workspace_bytes': (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    'stof_hilbert_f64': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                    _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_hilbert_streamed': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                         _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_gradpeak_moments': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_void_p, _c.c_int32,
                                         _c.c_void_p, _c.c_void_p]),
    'stof_gradpeak_blurred_stride': (_c.c_int64, [_c.c_int64, _c.c_int32]),
    'stof_gradpeak_moments_store': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_void_p, _c.c_int32,
                                               _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_grad_peak_detect_blurred': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_float,
                                                 _c.c_void_p, _c.c_int32, _c.c_int32, _c.c_int64, _c.c_void_p, _c.c_int64,
                                                 _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_gradpeak_threshold': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_grad_peak_detect': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_void_p, _c.c_int32,
                                         _c.c_float, _c.c_void_p, _c.c_int32, _c.c_int32, _c.c_int64, _c.c_void_p,
                                         _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_toa_detect_fused_ok': (_c.c_int, [_c.c_int64, _c.c_int32]),
    'stof_toa_detect': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_void_p, _c.c_int32, _c.c_float,
                                   _c.c_int32, _c.c_int32, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p,
                                   _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_toa_moments': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_void_p, _c.c_int32, _c.c_void_p,
                                    _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_iq2rf': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_double, _c.c_double, _c.c_double,
                              _c.c_int32, _c.c_void_p]),
    'stof_toa_rmse': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_float,
                                 _c.c_void_p, _c.c_void_p]),
    'stof_train_repack_floats': (_c.c_size_t, [_c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32]),
    'stof_train_repack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_void_p]),
    'stof_train_conv': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_void_p]),
    'stof_train_wgrad_workspace_bytes': (_c.c_size_t, [_c.c_int32, _c.c_int32, _c.c_int32]),
    'stof_train_wgrad': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_float, _c.c_int32, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_train_conv1': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p]),
    'stof_train_conv1_wgrad_workspace_bytes': (_c.c_size_t, []),
    'stof_train_conv1_wgrad': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_float, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_train_wgrad_batch_workspace_bytes': (_c.c_size_t, [_c.c_int32, _c.c_int32]),
    'stof_train_wgrad_batch': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_int64, _c.c_int32,
                                          _c.c_float, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_train_conv1_dgrad': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_float, _c.c_void_p]),
    'stof_train_sweep_blob_bytes': (_c.c_size_t, [_c.c_void_p]),
    'stof_train_sweep_pack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_train_sweep_dump_floats': (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    'stof_train_sweep': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p]),
    'stof_train_sweep_bwd_pack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_train_sweep_bwd': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p]),
    'stof_train_pool': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_void_p]),
    'stof_train_conv_last_dgrad': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_void_p]),
    'stof_train_sgb_dgrad_workspace_bytes': (_c.c_size_t, [_c.c_int32]),
    'stof_train_sgb_contract_dgrad': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                                 _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_train_sgb_wgrad_workspace_bytes': (_c.c_size_t, [_c.c_int32]),
    'stof_train_sgb_contract_wgrad': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                                 _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_float, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_train_pool_bwd': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_void_p]),
    'stof_train_sgb_blob_bytes': (_c.c_size_t, []),
    'stof_train_sgb_contract_pool': (_c.c_int, [_c.c_void_p] * 8 + [_c.c_int64, _c.c_int64, _c.c_void_p]),
    'stof_train_upsample_add': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_void_p]),
    'stof_train_upsample_add_c': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_void_p]),
    'stof_train_upsample_bwd': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_void_p]),
    'stof_train_loss': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_float, _c.c_float, _c.c_float, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_train_loss_target': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_train_loss_grad': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_float, _c.c_float, _c.c_float, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'stof_train_add': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p]),
    'stof_train_add_split': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p]),
    'stof_train_to_split_rows': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p]),
    'stof_train_add_split2': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p]),
    'stof_train_conv_last_dgrad_split': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_void_p]),
    'stof_train_sweep_split': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p]),
    'stof_train_sweep_bwd_split': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p]),
    'stof_train_wgrad_batch_split': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int32, _c.c_uint32, _c.c_uint32, _c.c_int64,
                                                _c.c_int64, _c.c_int32, _c.c_float, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'stof_train_adamw': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _c.c_int64, _c.c_void_p]),
    'stof_train_adamw_guarded': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StofnetLibraryMissing(
                f'{LIB_PATH} not found: build the HIP extension first (python -m stofnet_amd.build). '
                'stofnet_amd has no CPU/PyTorch fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def status_string(code: int) -> str:
    return lib().stof_status_string(int(code)).decode()


def check(code: int, what: str = ''):
    """Convert a stof_status into the exception class the reference would raise."""
    if code == STOF_OK:
        return
    msg = f'{what}: {status_string(code)}' if what else status_string(code)
    if code in (STOF_ERR_ODD_SGB_REMAINDER, STOF_ERR_CHANNELS, STOF_ERR_HIP, STOF_ERR_WORKSPACE, STOF_ERR_POOL_EMPTY):
        raise RuntimeError(msg)          # torch raises RuntimeError for shape mismatches
    if code == STOF_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise ValueError(msg)


def require_device(t: torch.Tensor, name: str = 'tensor') -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if t.device.type != 'cuda':
        raise RuntimeError(f'{name} is on {t.device}: stofnet_amd runs on a ROCm device only '
                           '(no CPU fallback; use oracle/ for a CPU check)')
    return t


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())
