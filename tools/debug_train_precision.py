#!/usr/bin/env python3
"""Diagnostic: per-parameter gradient error of the fp32 and f16x3 training modes against fp64 autograd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stofnet_amd import synth
from oracle import train_oracle as to
from stofnet_amd import StofNet
from stofnet_amd.training import StofNetTrainer
dev = torch.device('cuda:0')
r, sgs, L, n = 4, 80, 400, 3
sd = synth.synth_state_dict(r, seed=3, semi_global_scale=sgs)
x = synth.synth_echo(n, L, seed=11)
rng = np.random.default_rng(5)
gt = np.stack([np.sort(rng.integers(1, L * r, size=2)) for _ in range(n)])[:, None, :].astype(np.int64)
loss_ref, gref, pred_ref = to.loss_and_grads(sd, x, gt, r, sgs)
loss32, g32, _ = to.loss_and_grads(sd, x, gt, r, sgs, dtype=torch.float32)
res = {}
for prec in ('fp32', 'f16x3'):
    m = StofNet(upsample_factor=r, semi_global_scale=sgs)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr = StofNetTrainer(m.to(dev), precision=prec)
    loss, pred = tr.forward_backward(torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev))
    res[prec] = {k: tr.g[k].cpu().numpy().astype(np.float64) for k in tr.names}
    print(prec, 'loss rel err', abs(float(loss) - loss_ref) / loss_ref, 'pred', np.abs(pred.cpu().numpy() - pred_ref).max() / np.abs(pred_ref).max())
print(f"{'param':45s} {'hip fp32':>10s} {'hip f16x3':>10s} {'torch fp32':>10s}")
for k in gref:
    sc = np.abs(gref[k]).max()
    print(f'{k:45s} {np.abs(res["fp32"][k] - gref[k]).max() / sc:10.2e} {np.abs(res["f16x3"][k] - gref[k]).max() / sc:10.2e} {np.abs(g32[k] - gref[k]).max() / sc:10.2e}')
