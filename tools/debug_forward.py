"""GPU debugging aid: compare HIP forward stages with the oracle (not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import stofnet_oracle as so
from stofnet_amd import synth
from stofnet_amd import StofNet

dev = torch.device('cuda:0')
def run(sgs, r, L, N, seed=1, precision='fp32'):
    sd = synth.synth_state_dict(r, seed=seed, semi_global_scale=sgs)
    m = StofNet(upsample_factor=r, semi_global_scale=sgs, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    x = synth.synth_randn(N, L, seed=3)
    taps = {}
    ref = so.stofnet_forward(sd, x, r, sgs, taps=taps).numpy()
    y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    err = np.abs(y - ref)
    print(f'[{precision}] sgs={sgs} r={r} L={L} N={N}: max abs err {err.max():.3e} (ref max {np.abs(ref).max():.3e})')
    if err.max() > 1e-4 * np.abs(ref).max():
        for n in range(min(N, 3)):
            e = err[n, 0].reshape(L, r).max(1)
            bad = np.nonzero(e > 1e-4 * np.abs(ref).max())[0]
            print('  row', n, 'bad t count', bad.size, 'first', bad[:10], 'last', bad[-10:])
    if sgs != 1:
        P = L // 80
        ws = m._workspace.view(torch.float32)
        pooled = ws[:N * P * 512].reshape(N, P, 512).cpu().numpy()
        sgb = ws[N * P * 512: N * P * 576].reshape(N, P, 64).cpu().numpy()
        rp = taps['sgb_pooled'].numpy().transpose(0, 2, 1)
        re = taps['sgb_expand'].numpy().transpose(0, 2, 1)
        ep = np.abs(pooled - rp); ee = np.abs(sgb - re)
        print(f'  pooled err {ep.max():.3e} (max {np.abs(rp).max():.3e}); expand err {ee.max():.3e} (max {np.abs(re).max():.3e})')
        if ep.max() > 1e-4:
            bad = np.argwhere(ep > 1e-4)
            print('   pooled bad count', len(bad), 'examples', bad[:8].tolist(), 'windows', sorted(set(bad[:, 1].tolist()))[:30], 'oc', sorted(set(bad[:,2].tolist()))[:20])

for prec in ['fp32', 'f16x3']:
    run(1, 4, 400, 2, precision=prec)
    run(80, 4, 2000, 3, precision=prec)
    run(80, 10, 1536, 2, precision=prec)
