#!/usr/bin/env python3
"""Generate golden vectors by importing the reference (hahnec/stofnet) on CPU.

Run in the BUILD container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Emits data-only fixtures (inputs + expected outputs) into tests/golden/:
no reference source text is copied.  SURVEY.md §8c lists the cases (W, F1..F7).
Everything is produced with torch 2.10.0 CPU fp32, the reference's own modules
imported unmodified from /root/reference.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('STOFNET_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from stofnet_amd import synth  # noqa: E402  (our own deterministic input generator)

from models.stofnet import StofNet  # noqa: E402  (reference)
from models.gradpeak import GradPeak, toa_detect, grad_peak_detect, gaussian_kernel_1d, gaussian_filter_1d  # noqa: E402
from utils.sample_shuffle import SampleShuffle1D  # noqa: E402
from utils.hilbert import hilbert_transform, HilbertTransform  # noqa: E402
from utils.mask2samples import mask2coords, get_maxima_positions, coords2mask  # noqa: E402
from utils.metrics import toa_rmse  # noqa: E402
from utils.gaussian import gaussian_kernel  # noqa: E402

torch.manual_seed(3008)
torch.set_num_threads(os.cpu_count())

CKPTS = {
    'different-armadillo': ('different-armadillo-1439_rf-scale10_epoch_46.pth', 80),
    'graceful-snow': ('graceful-snow-1553_rf-scale20_epoch_52.pth', 80),
    'clean-serenity': ('clean-serenity-1656_rf-scale10_epoch_27.pth', 1),
}
manifest = {'torch': torch.__version__, 'numpy': np.__version__, 'cases': {}}


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    manifest['cases'][name] = {k: list(np.asarray(v).shape) for k, v in arrays.items()}
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def load_sd(key):
    fn, sgs = CKPTS[key]
    sd = torch.load(os.path.join(REF, 'ckpts', fn), map_location='cpu', weights_only=True)
    return sd, sgs


def ref_model(sd, r, sgs, conv_last=None):
    m = StofNet(upsample_factor=r, semi_global_scale=sgs).eval()
    sd = dict(sd)
    if conv_last is not None:
        sd['conv_last.weight'] = torch.from_numpy(conv_last[0])
        sd['conv_last.bias'] = torch.from_numpy(conv_last[1])
    m.load_state_dict(sd, strict=True)
    return m


def layer_taps(m, x):
    """Per-layer checkpoints via forward hooks on the reference module."""
    taps = {}
    hooks = []
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.Conv1d):
            hooks.append(mod.register_forward_hook(lambda _m, _i, o, name=name: taps.__setitem__(name, o.detach().clone())))
    if m.semi_global_block is not None:
        hooks.append(m.semi_global_block.register_forward_hook(lambda _m, _i, o: taps.__setitem__('x0', o.detach().clone())))
        hooks.append(m.semi_global_block.contract_pool.register_forward_hook(lambda _m, _i, o: taps.__setitem__('sgb_pooled', o.detach().clone())))
    with torch.no_grad():
        y = m(x)
    for h in hooks:
        h.remove()
    return y, taps


# ---------------------------------------------------------------- W: weights
for key in CKPTS:
    sd, _ = load_sd(key)
    save('weights_' + key, **{k: v.numpy() for k, v in sd.items()})

# ---------------------------------------------------------------- F1: forward maps
with torch.no_grad():
    sd_a, _ = load_sd('different-armadillo')
    sd_g, _ = load_sd('graceful-snow')
    sd_c, _ = load_sd('clean-serenity')

    # a) armadillo r=4, [8,1,2000]: 6 echo rows + 2 randn rows; per-layer taps for row 0
    x = np.concatenate([synth.synth_echo(6, 2000, seed=1), synth.synth_randn(2, 2000, seed=2)], 0)
    m = ref_model(sd_a, 4, 80)
    y, taps = layer_taps(m, torch.from_numpy(x))
    save('f1_armadillo_r4_L2000', x=x, y=y.numpy(),
         tap_conv1=taps['conv1'][0].numpy(), tap_sgb_pooled=taps['sgb_pooled'][0].numpy(),
         tap_sgb_expand=taps['semi_global_block.expand_conv'][0].numpy(),
         tap_x0=taps['x0'][0].numpy(), tap_conv3=taps['conv3'][0].numpy(),
         tap_conv12=taps['conv12'][0].numpy(),
         tap_conv_last=taps['conv_last'][0].numpy())

    # b) graceful-snow r=4, [4,1,1536] (PALA shape, L mod 80 = 16 -> pad 8+8)
    x = synth.synth_echo(4, 1536, seed=3)
    y = ref_model(sd_g, 4, 80)(torch.from_numpy(x))
    save('f1_snow_r4_L1536', x=x, y=y.numpy())

    # c) armadillo r=4, [2,1,20000]
    x = synth.synth_echo(2, 20000, seed=4)
    y = ref_model(sd_a, 4, 80)(torch.from_numpy(x))
    save('f1_armadillo_r4_L20000', x=x, y=y.numpy())

    # d) r=10: armadillo body + seeded conv_last (north-star shape)
    cl10 = synth.synth_conv_last(10, seed=10)
    x = synth.synth_echo(4, 2000, seed=5)
    y = ref_model(sd_a, 10, 80, cl10)(torch.from_numpy(x))
    save('f1_armadillo_r10_L2000', x=x, y=y.numpy(), conv_last_weight=cl10[0], conv_last_bias=cl10[1])

    # e) r=20: graceful-snow body + seeded conv_last
    cl20 = synth.synth_conv_last(20, seed=20)
    x = synth.synth_echo(2, 2000, seed=6)
    y = ref_model(sd_g, 20, 80, cl20)(torch.from_numpy(x))
    save('f1_snow_r20_L2000', x=x, y=y.numpy(), conv_last_weight=cl20[0], conv_last_bias=cl20[1])

    # f) no-SGB ablation (semi_global_scale=1), clean-serenity r=4
    x = synth.synth_echo(2, 2000, seed=7)
    y = ref_model(sd_c, 4, 1)(torch.from_numpy(x))
    save('f1_serenity_nosgb_r4_L2000', x=x, y=y.numpy())

    # g) fully seeded-random state dict (what bench.py uses at r=10), [2,1,2000]
    sdr = synth.synth_state_dict(10, seed=3008)
    m = StofNet(upsample_factor=10).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sdr.items()}, strict=True)
    x = synth.synth_randn(2, 2000, seed=8)
    y = m(torch.from_numpy(x))
    save('f1_seeded_r10_L2000', x=x, y=y.numpy())

    # h) arg-max onset indices for 1024 synthetic echoes (inputs regenerated from the seed)
    x = synth.synth_echo(1024, 2000, seed=11)
    m = ref_model(sd_a, 4, 80)
    ys = torch.cat([m(torch.from_numpy(x[i:i + 64])) for i in range(0, 1024, 64)], 0)
    idx = get_maxima_positions(ys, 20, None).numpy()
    coords = mask2coords(ys, window_size=20, threshold=None, upsample_factor=4).numpy()
    top2 = torch.topk(ys[:, 0], 2, dim=-1).values.numpy()
    save('f1_armadillo_r4_argmax1024', seed=11, indices=idx, coords=coords, top2=top2,
         ymax=ys.abs().amax(-1).numpy())
    # same maps, threshold mode th=0.015 (PALA setting, bash_scripts/array_pala_params.txt:1), first 16 rows
    idx_t = get_maxima_positions(ys[:16], 20, 0.015).numpy()
    coords_t = mask2coords(ys[:16], window_size=20, threshold=0.015, upsample_factor=4).numpy()
    save('f4_picker_on_maps', y=ys[:16].numpy(), idx_none=idx[idx[:, 0] < 16], idx_th=idx_t, coords_th=coords_t,
         coords_none=mask2coords(ys[:16], 20, None, 4).numpy(),
         coords_th_echo3=mask2coords(ys[:16], 20, 0.015, 4, echo_max=3).numpy(),
         coords_th_echo40=mask2coords(ys[:16], 20, 0.015, 4, echo_max=40).numpy())

# ---------------------------------------------------------------- F2: shuffle
arrs = {}
for r, c, w in [(4, 1, 50), (10, 1, 33), (20, 1, 17), (4, 16, 9), (3, 2, 5)]:
    xin = torch.arange(2 * r * c * w, dtype=torch.float32).reshape(2, r * c, w)
    arrs[f'in_r{r}_c{c}'] = xin.numpy()
    arrs[f'out_r{r}_c{c}'] = SampleShuffle1D(r)(xin).numpy()
save('f2_shuffle', **arrs)

# ---------------------------------------------------------------- F3: SGB length quirks
arrs, errs = {}, {}
with torch.no_grad():
    m = ref_model(sd_a, 4, 80)
    for L in [1536, 2000, 2040, 2578, 160, 96]:
        x = synth.synth_echo(1, L, seed=100 + L)
        arrs[f'x_L{L}'] = x
        arrs[f'y_L{L}'] = m(torch.from_numpy(x)).numpy()
    for L in [1999, 2001, 2041]:
        try:
            m(torch.from_numpy(synth.synth_echo(1, L, seed=100 + L)))
            errs[str(L)] = None
        except Exception as e:  # noqa: BLE001
            errs[str(L)] = [type(e).__name__, str(e)]
save('f3_sgb_lengths', **arrs)
manifest['f3_errors'] = errs

# ---------------------------------------------------------------- F4: picker hand cases
def hand_cases():
    cases = {}
    s = np.zeros((3, 1, 100), np.float32)
    s[0, 0, 10] = 2.0; s[0, 0, 50] = 2.0            # tie 10/50
    s[1, 0, 30] = 1.0; s[1, 0, 35] = 1.5            # 30 suppressed by 35 (window 21)
    s[2, 0, :] = -1.0                               # constant negative row
    cases['tie_nms_neg'] = s
    s = np.zeros((2, 1, 64), np.float32)
    s[0, 0, 0] = 3.0                                # index-0 hit
    s[1, 0, 20:25] = 1.0                            # plateau 20..24
    cases['idx0_plateau'] = s
    cases['empty'] = np.zeros((2, 1, 40), np.float32)
    s = np.zeros((2, 1, 200), np.float32)
    s[0, 0, [15, 60, 100, 150, 190]] = [1.0, 2.0, 1.6, 3.0, 0.4]
    s[1, 0, [5, 120]] = [1.7, 0.2]
    s[1, 0, 70:75] = -0.5
    cases['multi'] = s
    rng = np.random.default_rng(77)
    cases['noise'] = rng.standard_normal((4, 1, 500)).astype(np.float32)
    return cases


arrs = {}
shapes3d = {}
for name, s in hand_cases().items():
    arrs['in_' + name] = s
    for thn, th in [('none', None), ('zero', 0), ('1p5', 1.5), ('neg', -0.75)]:
        for em in [None, 3]:
            out = mask2coords(torch.from_numpy(s.copy()), window_size=20, threshold=th, upsample_factor=4, echo_max=em)
            arrs[f'out_{name}_th{thn}_em{em}'] = out.numpy()
        arrs[f'idx_{name}_th{thn}'] = get_maxima_positions(torch.from_numpy(s.copy()), 20, th).numpy()
    arrs[f'out_{name}_w5'] = mask2coords(torch.from_numpy(s.copy()), window_size=5, threshold=None, upsample_factor=1).numpy()
    arrs[f'out_{name}_w4_th'] = mask2coords(torch.from_numpy(s.copy()), window_size=4, threshold=0.5, upsample_factor=2).numpy()
save('f4_picker_hand', **arrs)

# ---------------------------------------------------------------- F5: Hilbert
arrs = {}
for n in [7, 16, 1536, 2000, 2001, 8000, 20000]:
    x = synth.synth_echo(2, n, seed=200 + n) if n >= 100 else synth.synth_randn(2, n, seed=200 + n)
    v = hilbert_transform(torch.from_numpy(x))
    arrs[f'x_n{n}'] = x
    arrs[f'env_n{n}'] = abs(v).numpy()
    if n <= 2001:
        arrs[f're_n{n}'] = v.real.numpy()
        arrs[f'im_n{n}'] = v.imag.numpy()
x = synth.synth_echo(2, 2000, seed=2200)
arrs['concat_in'] = x
arrs['concat_out'] = HilbertTransform(concat_oscil=True)(torch.from_numpy(x)).numpy()
save('f5_hilbert', **arrs)

# ---------------------------------------------------------------- F6: GradPeak
def gp_input(n_rows, L, seed, rf):
    """three clean-ish echoes (onsets 500/1203/800 scaled with rf/10) + noise rows"""
    x, on = synth.synth_echo(n_rows, L, seed=seed, noise=0.01, attack=3 * rf, tau=15.0 * rf,
                             carrier=0.2 / rf, return_onsets=True)
    return x, on


arrs = {}
gp_meta = {}
for rf in [10, 20]:
    x, on = gp_input(6, 2000, seed=300 + rf, rf=rf)
    arrs[f'x_rf{rf}'] = x
    arrs[f'onsets_rf{rf}'] = on
    xt = torch.from_numpy(x)
    env = abs(hilbert_transform(xt.squeeze(1)))
    g = rf // 6 * 5
    grad = gaussian_filter_1d(torch.gradient(env, spacing=g, dim=-1)[0], sigma=(g * 2 - 1) / 6)
    arrs[f'grad_rf{rf}'] = grad.numpy()
    arrs[f'taps_rf{rf}'] = gaussian_kernel_1d((g * 2 - 1) / 6).numpy()
    arrs[f'thdefault_rf{rf}'] = ((grad.std() ** 16) * 1.2e13).numpy()
    for thn, th in [('none', None), ('1em3', 1e-3), ('1em5', 1e-5)]:
        for oo in [True, False]:
            for emn, em in [('1', 1), ('inf', float('inf')), ('2', 2)]:
                key = f'out_rf{rf}_th{thn}_onset{int(oo)}_em{emn}'
                try:
                    out = GradPeak(threshold=th, rescale_factor=rf, echo_max=em, onset_opt=oo)(xt)
                    arrs[key] = out.numpy()
                    gp_meta[key] = 'ok'
                except Exception as e:  # noqa: BLE001
                    gp_meta[key] = [type(e).__name__, str(e)[:200]]
        try:
            arrs[f'echoes_rf{rf}_th{thn}'] = toa_detect(xt.squeeze(1), threshold=th, rescale_factor=rf).numpy()
        except Exception as e:  # noqa: BLE001
            gp_meta[f'echoes_rf{rf}_th{thn}'] = [type(e).__name__, str(e)[:200]]
# Q9 trigger: row 0 has edges but its only candidate gap (~500) is > 50*rf (rf=6 -> 300)
xq = np.zeros((2, 1, 800), np.float32)
xq[0, 0, 100:110] = np.linspace(0, 1, 10)
xq[0, 0, 110:600] = 1.0
xq[1] = synth.synth_echo(1, 800, seed=999)[0]
arrs['x_q9'] = xq
for rf, th in [(6, 1e-3), (1, 1e-2)]:
    key = f'q9_rf{rf}'
    try:
        out = GradPeak(threshold=th, rescale_factor=rf, echo_max=float('inf'), onset_opt=True)(torch.from_numpy(xq))
        arrs[key] = out.numpy()
        gp_meta[key] = 'ok'
    except Exception as e:  # noqa: BLE001
        gp_meta[key] = [type(e).__name__, str(e)[:200]]
# Q9 direct: crafted envelope, falling slope at 200 BEFORE the only rising slope at 600
tq = np.arange(800, dtype=np.float32)
envq = np.stack([1.0 - 1.0 / (1 + np.exp(-(tq - 200) / 8)) + 1.0 / (1 + np.exp(-(tq - 600) / 8)),
                 np.exp(-((tq - 300) / 40) ** 2)]).astype(np.float32)
arrs['env_q9'] = envq
try:
    out = grad_peak_detect(torch.from_numpy(envq), grad_step=5, threshold=1e-3, ival_smin=6, ival_smax=300)
    arrs['q9_direct'] = out.numpy()
    gp_meta['q9_direct'] = ['returned', list(out.shape)]
except Exception as e:  # noqa: BLE001
    gp_meta['q9_direct'] = [type(e).__name__, str(e)[:200]]
out = grad_peak_detect(torch.from_numpy(envq[1:]), grad_step=5, threshold=1e-3, ival_smin=6, ival_smax=300)
arrs['q9_direct_row1_only'] = out.numpy()
# no edges at all: all-zero input
try:
    out = GradPeak(threshold=1e-3, rescale_factor=10, echo_max=1, onset_opt=True)(torch.zeros(2, 1, 300))
    arrs['noedges'] = out.numpy()
    gp_meta['noedges'] = 'ok'
except Exception as e:  # noqa: BLE001
    gp_meta['noedges'] = [type(e).__name__, str(e)[:200]]
save('f6_gradpeak', **arrs)
manifest['f6_status'] = gp_meta

# ---------------------------------------------------------------- F7: host metrics
gt = torch.tensor([[10.0, 50.0, 0.0], [20.0, 0.0, 0.0], [0.0, 0.0, 0.0], [5.0, 7.0, 90.0]])
es = torch.tensor([[10.5, 80.0], [20.0, 21.0], [3.0, 0.0], [0.0, 0.0]])
arrs = {'gt': gt.numpy(), 'es': es.numpy()}
for tol in [1, 4]:
    arrs[f'rmse_tol{tol}'] = toa_rmse(gt, es, tol=tol).numpy()
arrs['gauss7'] = gaussian_kernel(7, 1.0)
arrs['gauss5_s2'] = gaussian_kernel(5, 2.0)
ref = torch.zeros(2, 1, 20)
samples = torch.tensor([[[3, 7, 0]], [[-2, 19, 5]]])
arrs['c2m_samples'] = samples.numpy()
arrs['c2m_mask'] = coords2mask(samples.clone(), ref).numpy()
save('f7_metrics', **arrs)

with open(os.path.join(HERE, 'manifest.json'), 'w') as f:
    json.dump(manifest, f, indent=1, sort_keys=True)
print('done')
