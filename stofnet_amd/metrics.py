"""toa_rmse (utils/metrics.py:9-41) on the gfx950 kernel: one wavefront per row, no python loop.
Device tensors only, like the rest of the package (the CPU restatement lives in oracle/pickers_oracle.py)."""
import torch

from . import _lib


def toa_rmse(gt_samples, es_samples, tol=1):
    """[N, ...] GT and estimate ToAs on a ROCm device -> [N, 7] = (rmse, precision, recall, jaccard, tp, fp, fn)."""
    _lib.require_device(gt_samples, 'gt_samples')
    n = gt_samples.shape[0]
    g = gt_samples.detach().reshape(n, -1).contiguous().float()
    e = es_samples.detach().to(g.device).reshape(n, -1).contiguous().float()
    out = torch.empty((n, 7), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().stof_toa_rmse(_lib.ptr(g), _lib.ptr(e), n, g.shape[1], e.shape[1], float(tol),
                                            _lib.ptr(out), _lib.stream_ptr(g.device)), 'stof_toa_rmse')
    return out


toa_rmse_device = toa_rmse
