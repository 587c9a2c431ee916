"""toa_rmse (utils/metrics.py:9-41).  `toa_rmse` runs on the gfx950 kernel for device tensors
(one wavefront per row, no python loop); CPU tensors take the host restatement below, which is
what `main.py`'s summary and the CPU tests use."""
import torch

from . import _lib


def _valid(x):
    return x[(x != 0) & (~torch.isnan(x)) & (~torch.isinf(x))]


def toa_rmse(gt_samples, es_samples, tol=1):
    if gt_samples.device.type == 'cuda':
        return toa_rmse_device(gt_samples, es_samples, tol)
    n = gt_samples.shape[0]
    mes, tps, fps, fns = (torch.zeros(n, device=gt_samples.device) for _ in range(4))
    for i in range(n):
        g = _valid(gt_samples[i].reshape(-1).float())
        e = _valid(es_samples[i].reshape(-1).float())
        if g.numel() == 0 or e.numel() == 0:
            continue
        mins = ((g[:, None] - e[None, :]) ** 2).min(-1).values
        hit = mins <= tol
        mes[i] = torch.mean(mins[hit]) ** .5
        tps[i] = hit.sum().float()
        fns[i] = (~hit).sum().float()
        fps[i] = e.numel() - tps[i]
    jaccards = tps / (fns + tps + fps) * 100
    precisions = tps / (fps + tps) * 100
    recalls = tps / (fns + tps) * 100
    return torch.stack([mes, precisions, recalls, jaccards, tps, fps, fns]).T


def toa_rmse_device(gt_samples, es_samples, tol=1):
    """[N, ...] GT and estimate ToAs on a ROCm device -> [N, 7] on the device."""
    _lib.require_device(gt_samples, 'gt_samples')
    n = gt_samples.shape[0]
    g = gt_samples.detach().reshape(n, -1).contiguous().float()
    e = es_samples.detach().to(g.device).reshape(n, -1).contiguous().float()
    out = torch.empty((n, 7), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().stof_toa_rmse(_lib.ptr(g), _lib.ptr(e), n, g.shape[1], e.shape[1], float(tol),
                                            _lib.ptr(out), _lib.stream_ptr(g.device)), 'stof_toa_rmse')
    return out
