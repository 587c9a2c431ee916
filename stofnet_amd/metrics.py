"""toa_rmse (utils/metrics.py:9-41): host-side metric over the ragged per-row ToA lists, kept
on the host as in the reference (a python loop over rows of tiny tensors)."""
import torch


def _valid(x):
    return x[(x != 0) & (~torch.isnan(x)) & (~torch.isinf(x))]


def toa_rmse(gt_samples, es_samples, tol=1):
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
