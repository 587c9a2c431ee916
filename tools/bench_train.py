#!/usr/bin/env python3
"""Time the training step (BASELINE.json configs[4], one GPU): fwd+bwd+AdamW at rf upsample 10, L=2000."""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stofnet_amd import synth  # input generator only
from stofnet_amd import StofNet
from stofnet_amd.training import StofNetTrainer

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--length', type=int, default=2000)
ap.add_argument('--r', type=int, default=10)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--warmup', type=int, default=2)
a = ap.parse_args()
dev = torch.device('cuda:0')
sd = synth.synth_state_dict(a.r, seed=1, semi_global_scale=80)
m = StofNet(upsample_factor=a.r, semi_global_scale=80)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev)
tr = StofNetTrainer(m)
x = torch.from_numpy(synth.synth_echo(a.batch, a.length, seed=4)).to(dev)
rng = np.random.default_rng(0)
gt = torch.from_numpy(np.sort(rng.integers(1, a.length * a.r, size=(a.batch, 1, 2)), -1)).to(dev)
for _ in range(a.warmup):
    tr.train_step(x, gt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(a.steps):
    loss, _ = tr.train_step(x, gt)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
# fwd 2*MACs; bwd = dgrad + wgrad = 2x fwd
from oracle.stofnet_oracle import flops_per_waveform
fl = 3 * flops_per_waveform(a.length, a.r) * a.batch
print(json.dumps({'train_step_ms': ms, 'waveforms_per_s': a.batch / ms * 1e3, 'batch': a.batch, 'L': a.length, 'r': a.r,
                  'tflops_fp32': fl / ms / 1e9, 'loss': float(loss)}))
