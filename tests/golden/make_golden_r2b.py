"""Golden vectors for the two baselines that ride on SampleShuffle1D, captured by importing
the reference on the CPU:  EDSR_1D (models/edsr_1d.py) with a reduced width / depth so that the fixture stays small, and
ESPCN_1D (models/espcn_1d.py) at its real size.  Stored: every parameter (by its state_dict name), a seeded input and
the reference's output.   python tests/golden/make_golden_r2b.py   (needs /root/reference; run in the build container)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
from models.edsr_1d import EDSR_1D      # noqa: E402
from models.espcn_1d import ESPCN_1D    # noqa: E402

out = {}
torch.manual_seed(1234)
for tag, model, shape in (('edsr_r4', EDSR_1D(num_channels=1, num_features=16, num_blocks=2, upscale_factor=4), (3, 1, 200)),
                          ('edsr_r2', EDSR_1D(num_channels=1, num_features=8, num_blocks=1, upscale_factor=2), (2, 1, 96)),
                          ('espcn_r4', ESPCN_1D(upscale_factor=4), (3, 1, 200)),
                          ('espcn_r10', ESPCN_1D(upscale_factor=10), (2, 1, 120))):
    model.eval()
    if tag.startswith('espcn'):                       # the reference's init leaves conv3 at std 0.001: make the output non-trivial
        with torch.no_grad():
            model.conv3.weight.mul_(300.0)
    x = torch.randn(*shape)
    with torch.no_grad():
        y = model(x)
    out[f'{tag}__x'] = x.numpy()
    out[f'{tag}__y'] = y.numpy()
    for k, v in model.state_dict().items():
        out[f'{tag}__p__{k}'] = v.numpy()
np.savez_compressed(os.path.join(HERE, 'f11_shuffle_riders.npz'), **out)
print({k: v.shape for k, v in out.items() if k.endswith(('__x', '__y'))})
