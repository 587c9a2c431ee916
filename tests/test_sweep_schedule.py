"""Validates the fused body kernel's sweep schedule (ring buffers, skewed layer
frontiers, in-place residual update, gap rows) against the oracle, on CPU."""
import numpy as np
import pytest
import torch

from conftest import load_weights
from oracle import stofnet_oracle as so
from stofnet_amd import synth
from oracle.sweep_emulator import sweep_forward


@pytest.mark.parametrize('L,r,S,RING,sgs', [(400, 4, 192, 256, 80), (336, 10, 64, 104, 80), (250, 4, 128, 168, 1)])
def test_sweep_matches_oracle(L, r, S, RING, sgs):
    p = synth.synth_state_dict(r, seed=5, semi_global_scale=sgs)
    x = synth.synth_randn(3, L, seed=9)
    taps = {}
    ref = so.stofnet_forward(p, x, r, sgs, torch.float64, taps=taps).numpy()
    sgb = taps['sgb_expand'].numpy() if sgs != 1 else None
    y = sweep_forward(p, x.astype(np.float64), sgb, r, S=S, RING=RING)
    assert np.abs(y - ref).max() < 1e-10
    # a work-group that owns only waveforms [1, 3) must reproduce them without row 0
    y2 = sweep_forward(p, x.astype(np.float64), sgb, r, S=S, RING=RING, n_begin=1, n_end=3)
    assert np.abs(y2 - ref[1:3]).max() < 1e-10
