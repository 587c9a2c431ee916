"""GPU time (events) and wall time per toa_detect call for the alternative launch sequences, per batch size:
explicit threshold: fused one-launch kernel vs envelope kernel + row kernel; default threshold: stof_toa_moments vs
envelope kernel + stof_gradpeak_moments.  -> gpurun_out/r04_gradpeak_paths.json"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import stofnet_amd.gradpeak as gp
from stofnet_amd import synth, toa_detect

L, rf = 2000, 10
out = []
for rows in (256, 1024, 2048, 4096, 8192, 32768):
    x = torch.from_numpy(synth.synth_echo(rows, L, seed=3, noise=0.01)).cuda()[:, 0].contiguous()
    for th, name, values in ((1e-3, 'explicit', {'fused': 1 << 62, 'envelope+rows': 0}),
                             (None, 'default', {'toa_moments': 1 << 62, 'envelope+moments': 0})):
        for label, v in values.items():
            gp._ONE_LAUNCH_MAX_ROWS = v
            for _ in range(3):
                toa_detect(x, threshold=th, rescale_factor=rf)
            torch.cuda.synchronize()
            reps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                toa_detect(x, threshold=th, rescale_factor=rf)
            e1.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e6
            rec = {'rows': rows, 'L': L, 'rf': rf, 'threshold': name, 'path': label, 'wall_us_per_call': round(wall, 1),
                   'event_us_per_call': round(e0.elapsed_time(e1) / reps * 1e3, 1)}
            print(json.dumps(rec), flush=True)
            out.append(rec)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/r04_gradpeak_paths.json', 'w'), indent=1)
