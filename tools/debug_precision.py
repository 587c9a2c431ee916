"""GPU debugging aid: error of each precision mode vs the reference golden maps and vs the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from conftest import golden, load_weights
from oracle import stofnet_oracle as so
from stofnet_amd import StofNet
dev = torch.device('cuda:0')
CASES = [('f1_armadillo_r4_L2000', 'different-armadillo', 4, 80), ('f1_snow_r4_L1536', 'graceful-snow', 4, 80),
         ('f1_armadillo_r10_L2000', 'different-armadillo', 10, 80), ('f1_snow_r20_L2000', 'graceful-snow', 20, 80),
         ('f1_serenity_nosgb_r4_L2000', 'clean-serenity', 4, 1)]
for case, wkey, r, sgs in CASES:
    g = golden(case); sd = load_weights(wkey)
    if 'conv_last_weight' in g.files:
        sd['conv_last.weight'], sd['conv_last.bias'] = g['conv_last_weight'], g['conv_last_bias']
    taps = {}
    y64 = so.stofnet_forward(sd, g['x'][:2], r, sgs, torch.float64, taps=taps).numpy()
    amax = {k: float(v.abs().max()) for k, v in taps.items()}
    ymax = np.abs(g['y']).max()
    print(case, 'ymax %.3g' % ymax, 'ref32 vs fp64: %.3e' % (np.abs(g['y'][:2] - y64).max() / ymax),
          'act max:', {k: round(v, 1) for k, v in amax.items() if k in ('conv1', 'x0', 'res3', 'res11', 'conv12', 'sgb_contract')})
    for prec in ['fp32', 'f16x3']:
        m = StofNet(upsample_factor=r, semi_global_scale=sgs, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.to(dev).eval()
        y = m(torch.from_numpy(g['x']).to(dev)).cpu().numpy()
        print(f'   {prec:6s} vs golden {np.abs(y - g["y"]).max() / ymax:.3e}   vs fp64 {np.abs(y[:2] - y64).max() / ymax:.3e}'
              f'   argmax equal {np.array_equal(y[:, 0].argmax(-1), g["y"][:, 0].argmax(-1))}')
