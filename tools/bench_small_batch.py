#!/usr/bin/env python3
"""Latency / throughput of StofNet.forward at small batches (segment mode of the body sweep)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stofnet_amd import synth
from stofnet_amd import StofNet
dev = torch.device('cuda:0')
r, L = 10, 2000
sd = synth.synth_state_dict(r, seed=3008)
for prec in ('fp32', 'f16x3'):
    for policy, name in ((1, 'unsegmented'), (0, 'auto')):
        m = StofNet(upsample_factor=r, precision=prec)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev).eval()
        m._seg_policy = policy
        for n in (1, 8, 64):
            x = torch.from_numpy(synth.synth_randn(n, L, seed=1)).to(dev)
            for _ in range(5):
                m(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(50):
                m(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 50
            print(json.dumps({'precision': prec, 'mode': name, 'batch': n, 'ms_per_forward': round(ms, 4), 'waveforms_per_s': round(n / ms * 1e3, 1)}))
# batches that do not divide the CU count evenly
for prec in ('f16x3',):
    for policy, name in ((1, 'unsegmented'), (0, 'auto')):
        m = StofNet(upsample_factor=r, precision=prec)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev).eval()
        m._seg_policy = policy
        for n in (300, 384, 700, 4096):
            x = torch.from_numpy(synth.synth_randn(n, L, seed=1)).to(dev)
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(10):
                m(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(json.dumps({'precision': prec, 'mode': name, 'batch': n, 'ms_per_forward': round(ms, 4), 'waveforms_per_s': round(n / ms * 1e3, 1)}))
