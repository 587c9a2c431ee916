"""CPU oracle of the training step (TEST INFRASTRUCTURE ONLY): the loss of main.py:226-232 restated
in torch and differentiated by autograd through oracle.stofnet_forward; AdamW via torch.optim."""
import numpy as np
import torch
import torch.nn.functional as F

from . import stofnet_oracle as so
from .pickers_oracle import gaussian_kernel


def blurred_target(pred_like, gt_true, kernel_size=7, sigma=1):
    """coords2mask + 7-tap blur (main.py:228-229), before the division by the maximum."""
    mask = torch.zeros_like(pred_like)
    idx = gt_true.clone()
    idx[idx < 0] = 0
    mask.scatter_(2, idx, 1)
    mask[..., :1] = 0
    k = torch.tensor(gaussian_kernel(kernel_size, sigma), dtype=pred_like.dtype)[None, None]
    return F.conv1d(mask, k, padding=kernel_size // 2)


def loss_fn(pred, gt_true, lambda_value=1e-2, mask_amplitude=20, kernel_size=7, sigma=1, blur_max=None):
    """pred [N,1,M]; gt_true [N,1,G] int64.  main.py:228-232 (coords2mask -> blur -> /max -> *20 -> MSE + lambda*L1).
    blur_max: the maximum to divide by when `pred` is only a shard of the batch (the reference's is batch-global)."""
    mask = torch.zeros_like(pred)
    idx = gt_true.clone()
    idx[idx < 0] = 0
    mask.scatter_(2, idx, 1)
    mask[..., :1] = 0
    k = torch.tensor(gaussian_kernel(kernel_size, sigma), dtype=pred.dtype)[None, None]
    blur = F.conv1d(mask, k, padding=kernel_size // 2)
    blur = blur / (blur.max() if blur_max is None else blur_max) * mask_amplitude
    return F.mse_loss(pred.squeeze(1), blur.squeeze(1)) + F.l1_loss(pred.squeeze(1), torch.zeros_like(pred.squeeze(1))) * lambda_value


def loss_and_grads(params, x, gt_true, r, sgs=80, dtype=torch.float64, **kw):
    p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items()}
    pred = so.stofnet_forward(p, torch.tensor(np.asarray(x), dtype=dtype), r, sgs, dtype)
    loss = loss_fn(pred, torch.as_tensor(gt_true), **kw)
    grads = torch.autograd.grad(loss, list(p.values()))
    return float(loss.detach()), {k: g.numpy() for k, g in zip(p.keys(), grads)}, pred.detach().numpy()
