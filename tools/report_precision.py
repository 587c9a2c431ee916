"""Measured distance of each MFMA mode from the reference's golden maps and from the float64 truth, per golden case.
Writes one JSON document (profiles/r03_precision.json); DESIGN.md section 3 quotes it and tests/test_gpu_parity.py takes
its f16x3-vs-reference tolerance from the worst case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from conftest import golden, load_weights
from oracle import stofnet_oracle as so
from stofnet_amd import StofNet

CASES = [
    ('f1_armadillo_r4_L2000', 'different-armadillo', 4, 80),
    ('f1_snow_r4_L1536', 'graceful-snow', 4, 80),
    ('f1_armadillo_r4_L20000', 'different-armadillo', 4, 80),
    ('f1_armadillo_r10_L2000', 'different-armadillo', 10, 80),
    ('f1_snow_r20_L2000', 'graceful-snow', 20, 80),
    ('f1_serenity_nosgb_r4_L2000', 'clean-serenity', 4, 1),
]
dev = torch.device('cuda:0')
rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(b).max())
out = {'unit': 'max |difference| / max |y|', 'cases': {}}
for case, wkey, r, sgs in CASES:
    g = golden(case)
    sd = load_weights(wkey)
    if 'conv_last_weight' in g.files:
        sd['conv_last.weight'], sd['conv_last.bias'] = g['conv_last_weight'], g['conv_last_bias']
    rows = min(2, g['x'].shape[0])
    truth = so.stofnet_forward(sd, g['x'][:rows], r, sgs, torch.float64, conv=so.conv1d_shifted_matmul).numpy()
    e = {'reference_fp32_vs_fp64_truth': rel(g['y'][:rows], truth)}
    for prec in ('fp32', 'f16x3'):
        m = StofNet(upsample_factor=r, semi_global_scale=sgs, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        y = m.to(dev).eval()(torch.from_numpy(g['x']).to(dev)).cpu().numpy()
        e[prec + '_vs_reference'] = rel(y, g['y'])
        e[prec + '_vs_fp64_truth'] = rel(y[:rows], truth)
        e[prec + '_argmax_equal'] = bool(np.array_equal(y[:, 0].argmax(-1), g['y'][:, 0].argmax(-1)))
    out['cases'][case] = e
    print(case, json.dumps(e), flush=True)
out['worst'] = {k: max(c[k] for c in out['cases'].values()) for k in
                ('fp32_vs_reference', 'f16x3_vs_reference', 'fp32_vs_fp64_truth', 'f16x3_vs_fp64_truth', 'reference_fp32_vs_fp64_truth')}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r03_precision.json'), 'w'), indent=1)
print(json.dumps(out['worst']))
