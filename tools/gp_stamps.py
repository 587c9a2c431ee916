"""Per-phase cycles of toa_fused_kernel (the run-time-plan fused kernel) from a -DSTOF_GP_STAMPS build
(SRC=gradpeak bash tools/build_variant.sh gp_stamps -DSTOF_GP_STAMPS):
    STOF_FUSED_CT=0 STOF_LIB_PATH=stofnet_amd/libstof_gp_stamps.so python tools/gp_stamps.py [rows L rf]
(STOF_FUSED_CT=0: row lengths with a compile-time plan otherwise take toa_fused_ct_kernel, which carries no stamps)"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from stofnet_amd import _lib, synth, toa_detect
import stofnet_amd.gradpeak as gp

gp._ONE_LAUNCH_MAX_ROWS = 1 << 62      # the fused kernel at any batch size

rows, L, rf = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 2000, 10)
x = torch.from_numpy(synth.synth_echo(rows, L, seed=1, noise=0.01)).cuda()[:, 0].contiguous()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
toa_detect(x, threshold=1e-3, rescale_factor=rf)
torch.cuda.synchronize()
lib.stof_debug_gp_stamps(buf, 1)
reps = 10
for _ in range(reps):
    toa_detect(x, threshold=1e-3, rescale_factor=rf)
torch.cuda.synchronize()
lib.stof_debug_gp_stamps(buf, 0)
names = ['load_pair', 'fft (analytic)', 'unmix -> envelope', 'stream + pair', 'finish_row', 'tables / wait for the group']
waves = rows * reps          # one stamping wave per row (waves 0 and 1 of each pair)
for n, v in zip(names, buf):
    print(f'{n:20s} {v / waves:10.0f} cycles per wave (100 MHz ticks x ?: s_memtime counts shader-clock-independent ticks)')
