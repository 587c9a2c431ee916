"""Times hilbert_envelope on float64 rows (stof_hilbert_f64) next to the fp32 kernel; prints one JSON line per shape."""
import json
import torch
from stofnet_amd.hilbert import hilbert_envelope

dev = torch.device('cuda:0')
for rows, n in [(4096, 2000), (1024, 20000), (512, 30720)]:
    out = {'rows': rows, 'n': n}
    for name, dt in (('f64', torch.float64), ('f32', torch.float32)):
        x = torch.randn(rows, n, device=dev, dtype=dt)
        for _ in range(3):
            hilbert_envelope(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            hilbert_envelope(x)
        b.record()
        torch.cuda.synchronize()
        out[name + '_us'] = round(a.elapsed_time(b) * 100, 1)
    print(json.dumps(out), flush=True)
