"""Pins the CPU oracle (oracle/) against golden vectors captured from the
reference (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden, load_weights
from oracle import pickers_oracle as po
from oracle import stofnet_oracle as so
from stofnet_amd import synth

MANIFEST = json.load(open(os.path.join(GOLDEN, 'manifest.json')))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# tolerance of SURVEY.md §8c: 1e-5 relative to max-abs(y) for maps
MAP_TOL = 1e-5

FWD_CASES = [
    ('f1_armadillo_r4_L2000', 'different-armadillo', 4, 80),
    ('f1_snow_r4_L1536', 'graceful-snow', 4, 80),
    ('f1_armadillo_r4_L20000', 'different-armadillo', 4, 80),
    ('f1_armadillo_r10_L2000', 'different-armadillo', 10, 80),
    ('f1_snow_r20_L2000', 'graceful-snow', 20, 80),
    ('f1_serenity_nosgb_r4_L2000', 'clean-serenity', 4, 1),
]


def params_for(case, wkey):
    p = load_weights(wkey)
    g = golden(case)
    if 'conv_last_weight' in g.files:
        p['conv_last.weight'] = g['conv_last_weight']
        p['conv_last.bias'] = g['conv_last_bias']
    return p, g


@pytest.mark.parametrize('case,wkey,r,sgs', FWD_CASES)
def test_forward_fp32_matches_reference(case, wkey, r, sgs):
    p, g = params_for(case, wkey)
    y = so.stofnet_forward(p, g['x'], r, sgs, torch.float32).numpy()
    assert y.shape == g['y'].shape
    assert rel_err(y, g['y']) < MAP_TOL


@pytest.mark.parametrize('case,wkey,r,sgs', FWD_CASES[:2] + FWD_CASES[3:])
def test_forward_fp64_shifted_matmul_matches_reference(case, wkey, r, sgs):
    """Independent conv spelling in float64: the reference fp32 sits within rounding of it."""
    p, g = params_for(case, wkey)
    y = so.stofnet_forward(p, g['x'][:2], r, sgs, torch.float64, conv=so.conv1d_shifted_matmul).numpy()
    assert rel_err(y, g['y'][:2]) < MAP_TOL


def test_forward_seeded_state_dict():
    g = golden('f1_seeded_r10_L2000')
    p = synth.synth_state_dict(10, seed=3008)
    y = so.stofnet_forward(p, g['x'], 10, 80).numpy()
    assert rel_err(y, g['y']) < MAP_TOL


def test_layer_taps():
    p, g = params_for('f1_armadillo_r4_L2000', 'different-armadillo')
    taps = {}
    so.stofnet_forward(p, g['x'][:1], 4, 80, taps=taps)
    for ours, theirs in [('conv1', 'tap_conv1'), ('sgb_pooled', 'tap_sgb_pooled'), ('x0', 'tap_x0'),
                         ('res3', 'tap_conv3'), ('conv12', 'tap_conv12'), ('conv_last', 'tap_conv_last')]:
        a = taps[ours][0].numpy()
        b = g[theirs]
        if ours == 'conv1':
            b = np.maximum(b, 0)          # hook sees the conv output before ReLU
        if ours in ('res3', 'conv12'):
            continue                      # hooks see the raw conv output, not the residual sum
        assert rel_err(a, b) < MAP_TOL, ours
    # the reference's expand_conv hook is pre-activation
    assert rel_err(taps['sgb_expand'][0].numpy(), np.where(g['tap_sgb_expand'] > 0, g['tap_sgb_expand'], 0.01 * g['tap_sgb_expand'])) < MAP_TOL


def test_argmax_indices_1024_rows_bit_exact():
    g = golden('f1_armadillo_r4_argmax1024')
    p = load_weights('different-armadillo')
    x = synth.synth_echo(1024, 2000, seed=int(g['seed']))
    ys = np.concatenate([so.stofnet_forward(p, x[i:i + 128], 4, 80).numpy() for i in range(0, 1024, 128)])
    am = ys[:, 0].argmax(-1)
    ref = g['indices']
    assert np.array_equal(np.bincount(ref[:, 0], minlength=1024), np.ones(1024, np.int64))
    margins = g['top2'][:, 0] - g['top2'][:, 1]
    bad = np.nonzero(am != ref[:, 1])[0]
    # report margin statistics (SURVEY §7 hard parts) and require exact indices
    assert bad.size == 0, f'{bad.size} flips; min margin {margins.min()}, margins at flips {margins[bad]}'
    assert np.array_equal(po.mask2coords(ys[:32], 20, None, 4), g['coords'][:32])


@pytest.mark.parametrize('L', [1536, 2000, 2040, 2578, 160, 96])
def test_sgb_length_quirk_ok(L):
    g = golden('f3_sgb_lengths')
    p = load_weights('different-armadillo')
    y = so.stofnet_forward(p, g[f'x_L{L}'], 4, 80).numpy()
    assert rel_err(y, g[f'y_L{L}']) < MAP_TOL


@pytest.mark.parametrize('L', [1999, 2001, 2041])
def test_sgb_odd_remainder_raises(L):
    assert MANIFEST['f3_errors'][str(L)][0] == 'RuntimeError'
    p = load_weights('different-armadillo')
    with pytest.raises(RuntimeError, match=r'must match the size of tensor b'):
        so.stofnet_forward(p, synth.synth_echo(1, L, seed=1), 4, 80)


def test_shuffle_bit_exact():
    g = golden('f2_shuffle')
    for r, c in [(4, 1), (10, 1), (20, 1), (4, 16), (3, 2)]:
        out = so.sample_shuffle(torch.from_numpy(g[f'in_r{r}_c{c}']), r).numpy()
        assert np.array_equal(out, g[f'out_r{r}_c{c}'])
    with pytest.raises(RuntimeError):
        so.sample_shuffle(torch.zeros(1, 7, 4), 3)


TH = {'none': None, 'zero': 0, '1p5': 1.5, 'neg': -0.75}


def test_picker_hand_cases_bit_exact():
    g = golden('f4_picker_hand')
    names = sorted({f[3:] for f in g.files if f.startswith('in_')})
    assert names
    for name in names:
        s = g['in_' + name]
        for thn, th in TH.items():
            assert np.array_equal(po.maxima_positions(s, 20, th), g[f'idx_{name}_th{thn}'].reshape(-1, 2)), (name, thn)
            for em in [None, 3]:
                exp = g[f'out_{name}_th{thn}_em{em}']
                out = po.mask2coords(s, 20, th, 4, em)
                assert out.shape == exp.shape
                if name == 'tie_nms_neg' and em == 3:
                    # row 2 is a constant row: every amplitude ties, and the reference's
                    # non-stable argsort(descending) (utils/mask2samples.py:125) picks an
                    # unspecified subset -> only the tie-free rows are pinned
                    out, exp = out[:2], exp[:2]
                assert np.array_equal(out, exp), (name, thn, em)
        assert np.array_equal(po.mask2coords(s, 5, None, 1), g[f'out_{name}_w5'])
        assert np.array_equal(po.mask2coords(s, 4, 0.5, 2), g[f'out_{name}_w4_th'])


def test_picker_on_network_maps():
    g = golden('f4_picker_on_maps')
    y = g['y']
    assert np.array_equal(po.maxima_positions(y, 20, None), g['idx_none'])
    assert np.array_equal(po.maxima_positions(y, 20, 0.015), g['idx_th'])
    assert np.array_equal(po.mask2coords(y, 20, 0.015, 4), g['coords_th'])
    assert np.array_equal(po.mask2coords(y, 20, None, 4), g['coords_none'])
    assert np.array_equal(po.mask2coords(y, 20, 0.015, 4, 3), g['coords_th_echo3'])
    assert np.array_equal(po.mask2coords(y, 20, 0.015, 4, 40), g['coords_th_echo40'])


@pytest.mark.parametrize('n', [7, 16, 1536, 2000, 2001, 8000, 20000])
def test_hilbert_envelope(n):
    g = golden('f5_hilbert')
    x = g[f'x_n{n}']
    v = po.hilbert_transform(x)
    assert np.abs(np.abs(v) - g[f'env_n{n}']).max() < 1e-5
    if n <= 2001:
        assert np.abs(v.real - g[f're_n{n}']).max() < 1e-5
        assert np.abs(v.imag - g[f'im_n{n}']).max() < 1e-5


def test_hilbert_odd_n_is_not_scipy():
    import scipy.signal
    x = golden('f5_hilbert')['x_n7']
    assert np.abs(np.abs(scipy.signal.hilbert(x, axis=-1)) - np.abs(po.hilbert_transform(x))).max() > 1e-2


def test_gradpeak_pieces():
    g = golden('f6_gradpeak')
    for rf in [10, 20]:
        gs = rf // 6 * 5
        assert np.allclose(po.gaussian_kernel_1d((gs * 2 - 1) / 6), g[f'taps_rf{rf}'], rtol=0, atol=1e-15)
        env = po.hilbert_envelope(g[f'x_rf{rf}'][:, 0])
        grad = po.smoothed_gradient(env, gs)
        assert np.abs(grad - g[f'grad_rf{rf}']).max() < 1e-6
        assert po.default_threshold(g[f'grad_rf{rf}']) == g[f'thdefault_rf{rf}']


@pytest.mark.parametrize('rf', [10, 20])
@pytest.mark.parametrize('thn,th', [('none', None), ('1em3', 1e-3), ('1em5', 1e-5)])
def test_gradpeak_outputs(rf, thn, th):
    g = golden('f6_gradpeak')
    x = g[f'x_rf{rf}']
    exp_e = g[f'echoes_rf{rf}_th{thn}']
    got_e = po.toa_detect(x[:, 0], th, rf)
    assert got_e.shape == exp_e.shape
    assert np.array_equal(got_e[..., :2], exp_e[..., :2])
    assert np.abs(got_e[..., 2] - exp_e[..., 2]).max() < 1e-5
    for oo in [True, False]:
        for emn, em in [('1', 1), ('inf', float('inf')), ('2', 2)]:
            key = f'out_rf{rf}_th{thn}_onset{int(oo)}_em{emn}'
            assert MANIFEST['f6_status'][key] == 'ok'
            out = po.gradpeak_forward(x, th, rf, em, oo)
            assert np.array_equal(out, g[key]), key


def test_gradpeak_q9_degenerate_is_pinned_as_raises():
    g = golden('f6_gradpeak')
    assert MANIFEST['f6_status']['q9_direct'] == ['returned', [3, 0]]
    with pytest.raises(po.GradPeakDegenerate):
        po.grad_peak_detect(g['env_q9'], grad_step=5, threshold=1e-3, ival_smin=6, ival_smax=300)
    out = po.grad_peak_detect(g['env_q9'][1:], grad_step=5, threshold=1e-3, ival_smin=6, ival_smax=300)
    assert np.array_equal(out[..., :2], g['q9_direct_row1_only'][..., :2])


def test_host_metrics():
    g = golden('f7_metrics')
    for tol in [1, 4]:
        exp = g[f'rmse_tol{tol}']
        got = po.toa_rmse(g['gt'], g['es'], tol)
        assert np.allclose(got, exp, rtol=1e-6, atol=0, equal_nan=True)
    assert np.allclose(po.gaussian_kernel(7, 1.0), g['gauss7'], rtol=0, atol=1e-16)
    assert np.allclose(po.gaussian_kernel(5, 2.0), g['gauss5_s2'], rtol=0, atol=1e-16)
    assert np.array_equal(po.coords2mask(g['c2m_samples'], (2, 1, 20)), g['c2m_mask'])


def test_flops_model():
    assert abs(so.flops_per_waveform(2000, 4) - 1.9305e9) < 1e6
    assert abs(so.flops_per_waveform(2000, 10) - 1.9351e9) < 1e6
    assert abs(so.flops_per_waveform(1536, 4) - 1.4826e9) < 1e6
    assert abs(so.flops_per_waveform(2000, 4, 1) - 1.2669e9) < 1e6


def test_f8_training_oracle_matches_reference_step():
    """oracle/train_oracle.py (loss of main.py:228-232 + autograd) against the reference's own fwd+bwd (f8)."""
    from oracle import train_oracle as to
    g = golden('f8_training')
    sd = load_weights('different-armadillo')
    loss, grads, pred = to.loss_and_grads(sd, g['frame'], g['gt_true'], 4, 80, dtype=torch.float32)
    assert np.abs(pred - g['masks_pred']).max() <= 1e-5 * np.abs(g['masks_pred']).max()
    assert abs(loss - float(g['loss0'])) < 1e-6 * float(g['loss0'])
    for k, v in grads.items():
        ref = g['grad.' + k]
        assert np.abs(v - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-9, k


# ---------------------------------------------------------------- round-2 fixtures (tests/golden/make_golden_r2.py)
def test_iq2rf_oracle_pinned_by_reference():
    """f2 row of SURVEY 8f: the oracle's iq2rf against the reference's own ChirpDataset.iq2rf (datasets/chirp_dataset.py:80-91)."""
    g = golden('f10_iq2rf')
    for rf in (1, 2.5, 10, 20):
        got = po.iq2rf(g['iq'], float(g['fc']), float(g['fs']), rf, normalize=False)
        assert np.abs(got - g[f'rf_{rf}']).max() < 1e-12


def test_short_rows_status_is_recorded():
    import json
    import os
    from conftest import GOLDEN
    st = json.load(open(os.path.join(GOLDEN, 'manifest_r2.json')))['f3_short_status']
    assert st['40'][0] == st['78'][0] == st['79'][0] == 'RuntimeError' and 'max_pool1d' in st['79'][1]
    assert st['80'] == ['ok', [2, 1, 320]] and st['82'] == ['ok', [2, 1, 328]]
    g = golden('f3_short_lengths')
    sd = load_weights('different-armadillo')
    for L in (80, 82):
        y = so.stofnet_forward(sd, g[f'x_L{L}'], 4, 80).numpy()
        assert np.abs(y - g[f'y_L{L}']).max() < 1e-5 * np.abs(g[f'y_L{L}']).max()


def test_long_rows_oracle_envelope_and_gradpeak():
    g = golden('f9_long_rows')
    L, rows, seed = 30720, int(g['rows_L30720']), int(g['seed_L30720'])
    x = synth.synth_echo(rows, L, seed=seed, noise=0.0005, attack=300, tau=3000.0, carrier=0.001)[:, 0]
    env = po.hilbert_envelope(x)
    assert np.abs(env[:, ::7] - g['env_L30720']).max() < 1e-5
    got = po.toa_detect(x, 1e-4, 20, env=env)
    assert np.array_equal(got[..., :2], g['idx_L30720_th1em4'])


def test_gradpeak_1024_oracle_agrees_where_clear_of_the_threshold():
    """The float64 oracle against the reference's own 1024-row result: identical wherever the smoothed gradient stays
    clear of the thresholds (a crossing is one float comparison; fp64 and fp32 round it differently)."""
    g = golden('f9_gradpeak_1024')
    x = synth.synth_echo(1024, 2000, seed=int(g['seed_rf10']), noise=0.01)[:256, 0]
    env = po.hilbert_envelope(x)
    sm = po.smoothed_gradient(env, 5)
    got = po.toa_detect(x, 1e-3, 10, env=env)
    exp = g['idx_rf10_th1em3'][:256]
    k = max(got.shape[1], exp.shape[1])
    pad = lambda a: np.pad(a, ((0, 0), (0, k - a.shape[1]), (0, 0)))
    same = (pad(got[..., :2]) == pad(exp)).all(axis=(1, 2))
    clear = (np.minimum(np.abs(sm - 1e-3), np.abs(sm + 2.5e-4)) > 2e-6 * np.abs(sm).max()).all(axis=1)
    assert same[clear].all() and clear.mean() > 0.8


def test_argmax4096_oracle_subset():
    g = golden('f1_armadillo_r4_argmax4096')
    x = synth.synth_echo(4096, 2000, seed=int(g['seed']))[:48]
    y = so.stofnet_forward(load_weights('different-armadillo'), x, 4, 80).numpy()
    assert np.array_equal(y[:, 0].argmax(-1), g['indices'][:48])


# ---------------------------------------------------------------- round 3 fixtures (tests/golden/make_golden_r3.py)
def test_argmax4096_r10_oracle_subset():
    g = golden('f1_armadillo_r10_argmax4096')
    sd = load_weights('different-armadillo')
    sd['conv_last.weight'], sd['conv_last.bias'] = synth.synth_conv_last(10, seed=int(g['conv_last_seed']))
    x = synth.synth_echo(4096, 2000, seed=int(g['seed']))[:32]
    y = so.stofnet_forward(sd, x, 10, 80).numpy()
    assert np.array_equal(y[:, 0].argmax(-1), g['indices'][:32])
    top2 = np.sort(y[:, 0], -1)[:, -2:][:, ::-1]
    assert np.abs(top2 - g['top2'][:32]).max() < 1e-5 * np.abs(g['top2']).max()


def sgb_input(n, L, seed):
    return np.random.default_rng(seed).standard_normal((n, 64, L)).astype(np.float32)


@pytest.mark.parametrize('L', [2000, 1536, 160])
def test_semi_global_block_standalone_oracle(L):
    """SemiGlobalBlock.forward alone (models/stofnet.py:98-117) with the checkpoint's block."""
    import torch
    g = golden('f13_sgb_standalone')
    sd = load_weights('different-armadillo')
    y = so.semi_global_block(torch.from_numpy(sgb_input(2, L, 1300 + L)), sd, 'semi_global_block.', 80, torch.float32).numpy()
    assert np.abs(y[..., ::5] - g[f'y80_L{L}']).max() < 1e-5 * np.abs(g[f'y80_L{L}']).max()
    assert int(g['err80_L2001']) == 1
    with pytest.raises(so.OddSemiGlobalRemainder):
        so.semi_global_block(torch.from_numpy(sgb_input(1, 2001, 1)), sd, 'semi_global_block.', 80, torch.float32)


@pytest.mark.parametrize('scale', [40, 20])
def test_other_semi_global_scales_oracle(scale):
    """models/stofnet.py:11 accepts any semi_global_scale: feat_scale = scale // 10, pool / upsample by scale."""
    import torch
    g = golden('f13_sgb_standalone')
    sd = synth.synth_state_dict(4, seed=3000 + scale, semi_global_scale=scale)
    assert sd['semi_global_block.contract_conv.weight'].shape[0] == 64 * (scale // 10)
    for L in (2000, 1536):
        y = so.stofnet_forward(sd, synth.synth_echo(2, L, seed=scale + L), 4, scale).numpy()
        assert np.abs(y - g[f'net{scale}_y_L{L}']).max() < 1e-5 * np.abs(g[f'net{scale}_y_L{L}']).max()
    yb = so.semi_global_block(torch.from_numpy(sgb_input(2, 1000, 1300 + scale)), sd, 'semi_global_block.', scale, torch.float32).numpy()
    assert np.abs(yb[..., ::5] - g[f'y{scale}_L1000']).max() < 1e-5 * np.abs(g[f'y{scale}_L1000']).max()


def test_pala_gradpeak_oracle_end_to_end():
    """The PALA GradPeak configuration (main.py:163-164: echo_max inf, peak column; main.py:301 flattening; rf 20) on a
    synthetic [B, C, S] frame stack: peaks and onsets of the reference."""
    g = golden('f12_pala_gradpeak')
    B, C, S, seed = (int(g[k]) for k in ('B', 'C', 'S', 'seed'))
    x = synth.pala_frames(B, C, S, seed).reshape(-1, 1, S)
    env = po.hilbert_envelope(x[:, 0])
    for thn, th in (('1em5', 1e-5), ('1em4', 1e-4)):
        peaks = po.gradpeak_forward(x, th, 20, float('inf'), False, env=env)
        onsets = po.gradpeak_forward(x, th, 20, float('inf'), True, env=env)
        assert np.array_equal(peaks, g[f'peaks_th{thn}']) and np.array_equal(onsets, g[f'onsets_th{thn}'])


@pytest.mark.parametrize('rf', [10, 20])
@pytest.mark.parametrize('tag,th', [('th1e-3', 1e-3), ('thdef', None)])
def test_gradpeak_oracle_matches_reference_on_4096_rows(rf, tag, th):
    """f9_gradpeak_4096 (make_golden_r4.py): the float64 oracle and the reference's fp32 toa_detect pick the same onset / peak
    samples on every one of the 4096 rows (so the GPU test may demand the same)."""
    from stofnet_amd import synth
    g = np.load(os.path.join(GOLDEN, 'f9_gradpeak_4096.npz'))
    if rf == 20 and th is None:
        rows = 1024                       # the default-threshold case at rf 20 walks 18 echoes per row: bound the CPU time
    else:
        rows = 2048
    x = synth.synth_echo(int(g['rows']), int(g['L']), seed=int(g['seed']), noise=float(g['noise']))[:, 0]
    ref = g[f'rf{rf}_{tag}']
    if th is None:
        got = po.toa_detect(x, th, rf)    # the default threshold is a statistic of the WHOLE batch: all rows go in
        rows = x.shape[0]
    else:
        got = po.toa_detect(x[:rows], th, rf)
    k = max(got.shape[1], ref.shape[1])
    pad = lambda a: np.pad(a, ((0, 0), (0, k - a.shape[1]), (0, 0)))
    assert np.array_equal(pad(got)[..., :2], pad(ref[:rows])[..., :2])
