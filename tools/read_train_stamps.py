"""Diagnostic (STOF_STAMPS build, STOF_LIB_PATH=stofnet_amd/libstof_NAME.so): per-segment cycle sums of the TRAINING sweeps
(forward with dumps, backward) over one C5 step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stofnet_amd import synth, StofNet, _lib
from stofnet_amd.training import StofNetTrainer
dev = torch.device('cuda:0')
n, L, r = 256, 2000, 10
m = StofNet(upsample_factor=r)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(r, seed=3).items()})
tr = StofNetTrainer(m.to(dev))
x = torch.from_numpy(synth.synth_echo(n, L, seed=2)).to(dev)
gt = torch.randint(1, L * r, (n, 1, 2), generator=torch.Generator().manual_seed(1)).sort(-1).values.to(dev)
for _ in range(3):
    tr.train_step(x, gt)
torch.cuda.synchronize()
lib = _lib.lib()
lib.stof_debug_train_stamps.argtypes = [ctypes.c_int, ctypes.c_void_p]
names = ['raw+bar', 'in pass', 'barrier', 'layer setup', 'both passes', 'exposed epilogue']
for which, nm in ((0, 'forward (DUMP)'), (1, 'backward')):
    st = np.zeros((256, 4, 8), dtype=np.uint64)
    assert lib.stof_debug_train_stamps(which, st.ctypes.data) == 0
    st = st.astype(np.float64)
    tot, steps = st[:, :, 6].mean(), st[0, 0, 7]
    print(f'[{nm}] cycles/wave {tot:.5g}, steps of wg0 {steps:.0f} (all wgs: {np.unique(st[:, 0, 7])}), cycles/step {tot / steps:.0f}')
    for i, nme in enumerate(names):
        print(f'  {nme:18s} {st[:, :, i].mean() / tot * 100:5.1f} %   per step {st[:, :, i].mean() / steps:8.0f}')
    print('  accounted', st[:, :, :6].sum(-1).mean() / tot * 100)
