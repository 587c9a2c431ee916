"""Profiling target: Hilbert envelope + GradPeak kernels only (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stofnet_amd import synth
from stofnet_amd.hilbert import hilbert_envelope
from stofnet_amd.gradpeak import toa_detect

dev = torch.device('cuda:0')
reps = int(os.environ.get('REPS', '5'))
only = os.environ.get('ONLY')
for n, rows in ((2000, 4096), (8000, 4096), (20000, 1024)):
    if only and int(only) != n:
        continue
    x = torch.from_numpy(synth.synth_randn(rows, n, seed=1)).to(dev)[:, 0].contiguous()
    for _ in range(reps):
        hilbert_envelope(x)
x = torch.from_numpy(synth.synth_echo(4096, 2000, seed=3, noise=0.01)).to(dev)[:, 0].contiguous()
for _ in range(0 if only else reps):
    toa_detect(x, 1e-3, 10)
torch.cuda.synchronize()
