"""toa_detect on the 4096-row reference golden (tests/golden/f9_gradpeak_4096) through both launch sequences: rows that differ
from the reference, and the wall time per call."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stofnet_amd import gradpeak as gp, synth, toa_detect
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'f9_gradpeak_4096.npz'))
x = torch.from_numpy(synth.synth_echo(int(g['rows']), int(g['L']), seed=int(g['seed']), noise=float(g['noise']))[:, 0]).cuda()
out = []
for name, limit in (('envelope kernel + row kernels (rows > 3072)', 3072), ('fused kernels at every batch size', 1 << 30)):
    gp._ONE_LAUNCH_MAX_ROWS = limit
    for rf in (10, 20):
        for tag, th in (('th1e-3', 1e-3), ('thdef', None)):
            got = toa_detect(x, threshold=th, rescale_factor=rf).cpu().numpy()
            ref = g[f'rf{rf}_{tag}']
            diff = (got[..., :2] != ref[..., :2]).any(axis=(1, 2)) if got.shape == ref.shape else np.ones(len(ref), bool)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): toa_detect(x, threshold=th, rescale_factor=rf)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            out.append({'path': name, 'rf': rf, 'threshold': tag, 'rows_differing_from_reference': int(diff.sum()), 'rows': np.nonzero(diff)[0][:8].tolist(), 'wall_us_per_call': round(dt * 1e6, 1)})
            print(json.dumps(out[-1]), flush=True)
