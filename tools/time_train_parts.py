#!/usr/bin/env python3
"""Where does the training step go?  Forward (activations kept), loss, backward, AdamW, timed separately with device events."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stofnet_amd import StofNet, synth
from stofnet_amd.training import StofNetTrainer

prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
nb, L, r = 256, 2000, 10
dev = torch.device('cuda:0')
sd = synth.synth_state_dict(r, seed=3008)
m = StofNet(upsample_factor=r)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
tr = StofNetTrainer(m.to(dev), precision=prec)
x = torch.from_numpy(synth.synth_echo(nb, L, seed=3008)).to(dev)
gt = torch.from_numpy(np.sort(np.random.default_rng(0).integers(1, L * r, size=(nb, 1, 2)), -1)).to(dev)
for _ in range(3):
    tr.train_step(x, gt)
torch.cuda.synchronize()
ev = [torch.cuda.Event(True) for _ in range(4)]
acc = np.zeros(3)
reps = 10
for _ in range(reps):
    ev[0].record()
    pred, saved = tr._forward_saved(tr.p, x)
    ev[1].record()
    n, mm = pred.shape
    target, dpred = torch.empty_like(pred), torch.empty_like(pred)
    tmax = torch.empty(1, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float64, device=dev)
    tr._loss_kernels(pred, gt.reshape(n, -1).contiguous(), n, mm, 2.0 ** 19, target, tmax, dpred, loss)
    ev[2].record()
    tr._backward_saved(saved, dpred, tr._grad_views, 2.0 ** 19)
    tr.step()
    ev[3].record()
    torch.cuda.synchronize()
    acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
print(json.dumps({'precision': prec, 'batch': nb, 'forward_ms': acc[0] / reps, 'loss_ms': acc[1] / reps, 'backward_adamw_ms': acc[2] / reps}))
