"""Dataset-side preparation of the chirp RF input on the GPU:
`iq2rf` = ChirpDataset.iq2rf (datasets/chirp_dataset.py:80-91) fused with NormalizeVol
(utils/transforms.py:13).  Today the reference does this per sample in numpy float64 inside
DataLoader workers; here `rf_scale_factor` becomes a device-side knob."""
import torch

from . import _lib


def iq2rf(iq_data: torch.Tensor, fc: float, fs: float, rescale_factor=1, normalize: bool = True) -> torch.Tensor:
    """iq_data: complex64 [N, len] or float32 [N, len, 2] on a ROCm device -> float32 [N, int(len*rescale_factor)]."""
    _lib.require_device(iq_data, 'iq_data')
    if iq_data.is_complex():
        iq = torch.view_as_real(iq_data.to(torch.complex64).contiguous())
    else:
        iq = iq_data.float().contiguous()
    if iq.dim() != 3 or iq.shape[-1] != 2:
        raise RuntimeError('iq_data must be complex [N, len] or real [N, len, 2]')
    n, ln, _ = iq.shape
    m = int(ln * rescale_factor)
    rf = torch.empty((n, m), dtype=torch.float32, device=iq.device)
    with torch.cuda.device(iq.device):
        _lib.check(_lib.lib().stof_iq2rf(_lib.ptr(iq), _lib.ptr(rf), n, ln, float(rescale_factor), float(fc),
                                         float(fs), 1 if normalize else 0, _lib.stream_ptr(iq.device)), 'stof_iq2rf')
    return rf
