"""Diagnostic (STOF_STAMPS build only): per-segment cycle shares of the body sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stofnet_amd import synth
from stofnet_amd import StofNet
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
N, L, r = 4096, 2000, 10
sd = synth.synth_state_dict(r, seed=3008)
m = StofNet(upsample_factor=r, precision=prec)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev).eval()
x = torch.from_numpy(synth.synth_randn(N, L, seed=3008)).to(dev)
for _ in range(3):
    y = m(x)
torch.cuda.synchronize()
P = L // 80
off = N * P * 576 * 4 + 256
st = m._workspace[off:off + 256 * 4 * 8 * 8].view(torch.int64).reshape(256, 4, 8).cpu().numpy().astype(np.float64)
names = ['raw+bar', 'x0 pass', 'barrier', 'layer setup', 'chunk loop', 'epilogue']
tot = st[:, :, 6].mean()
print(f'[{prec}] total cycles/wave {tot:.4g}, steps {st[0,0,7]:.0f}, cycles/step {tot / st[0,0,7]:.0f}')
for i, nme in enumerate(names):
    print(f'  {nme:12s} {st[:, :, i].mean() / tot * 100:5.1f} %   (per step {st[:, :, i].mean() / st[0,0,7]:.0f} cyc; wave spread {st[:, :, i].mean(0) / tot * 100})')
print('  accounted', st[:, :, :6].sum(-1).mean() / tot * 100)
