"""StofNet built with constructor arguments other than the shipped ones (models/stofnet.py:11): num_blocks, body kernel size,
semi_global_scale, r.  Golden `f14_ctor_variants` holds the reference's own forward result and autograd gradients
(tests/golden/make_golden_r4b.py); parameters and inputs come from numpy seeds (tests/ctor_variants.py)."""
import numpy as np
import pytest
import torch

from conftest import golden
from ctor_variants import VARIANTS, SGB_VARIANTS, sgb_input, variant_params, variant_input, variant_cotangent
from oracle import stofnet_oracle as so


def _case(name):
    from stofnet_amd import StofNet
    var, g = VARIANTS[name], golden('f14_ctor_variants')
    seed = int(g[f'{name}.seed'])
    m = StofNet(**var['ctor'])
    shapes = {n: tuple(t.shape) for n, t in m.state_dict().items()}
    params = variant_params(shapes, seed)
    x = variant_input(var['N'], var['L'], seed)
    t = variant_cotangent(var['N'], var['L'] * var['ctor']['upsample_factor'], seed)
    grads = {k[len(name) + 6:]: g[k] for k in g.files if k.startswith(name + '.grad.')}
    return var, m, params, x, t, g[f'{name}.y'], g[f'{name}.dx'], grads


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('name', list(VARIANTS))
def test_oracle_forward_and_gradients_match_reference(name):
    """Pins the generalised oracle (num_blocks / body kernel read from the state_dict) on the reference's output."""
    var, m, params, x, t, y_ref, dx_ref, grads_ref = _case(name)
    c = var['ctor']
    assert set(params) == set(m.state_dict())          # same state_dict names as the reference's constructor gave
    y = so.stofnet_forward(params, x, c['upsample_factor'], c['semi_global_scale'], torch.float32)
    assert y.shape == y_ref.shape
    assert rel(y.numpy(), y_ref) < 2e-6
    p64 = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    (so.stofnet_forward(p64, x64, c['upsample_factor'], c['semi_global_scale'], torch.float64) * torch.from_numpy(t).double()).sum().backward()
    assert rel(x64.grad.numpy(), dx_ref) < 2e-5
    assert grads_ref
    for n, gr in grads_ref.items():
        assert rel(p64[n].grad.numpy(), gr) < 2e-5, n


def test_constructor_accepts_what_the_kernels_serve():
    from stofnet_amd import StofNet
    assert StofNet(num_blocks=13)._fused_sweep() and StofNet(num_blocks=13)._supported()
    for var in VARIANTS.values():
        m = StofNet(**var['ctor'])
        assert m._supported() and not m._fused_sweep()
        assert m.residual_layers == list(range(3, m.num_blocks - 1, 2)) + [m.num_blocks - 1, m.num_blocks]
    assert not StofNet(num_features=32)._supported()
    assert not StofNet(kernel_sizes=[9, 9, 3])._supported()
    assert not StofNet(num_blocks=3)._supported()      # models/stofnet.py:60 fails for it too
    assert not StofNet(in_channels=2)._supported()
    with pytest.raises(ValueError):
        so.stofnet_forward({'conv1.weight': np.zeros((64, 1, 9), np.float32), 'conv2.weight': np.zeros((64, 64, 7), np.float32)},
                           np.zeros((1, 1, 8), np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('f16x3', 2e-5)])
@pytest.mark.parametrize('name', list(VARIANTS))
def test_gpu_forward_matches_reference(name, precision, tol):
    """Inference of the variants runs layer by layer on the channel-last MFMA kernels (no torch convolution on the path)."""
    var, m, params, x, t, y_ref, _, _ = _case(name)
    m.precision = precision
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to('cuda:0').eval()
    with torch.no_grad():
        y = m(torch.from_numpy(x).to('cuda:0'))
    assert y.shape == y_ref.shape and y.dtype == torch.float32
    assert rel(y.cpu().numpy(), y_ref) < tol


@pytest.mark.gpu
@pytest.mark.parametrize('tp,tol', [('fp32', 2e-4), ('f16x3', 2e-3)])
@pytest.mark.parametrize('name', list(VARIANTS))
def test_gpu_gradients_match_reference_autograd(name, tp, tol):
    """Train-mode forward + backward through the autograd boundary: every kept parameter gradient and d loss / d x against
    the reference's autograd (tolerances of tests/test_gpu_training.py)."""
    from stofnet_amd import StofNet
    var, _, params, x, t, y_ref, dx_ref, grads_ref = _case(name)
    m = StofNet(**var['ctor'], train_precision=tp)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to('cuda:0').train()
    xg = torch.from_numpy(x).to('cuda:0').requires_grad_()
    y = m(xg)
    assert rel(y.detach().cpu().numpy(), y_ref) < 2e-5
    (y * torch.from_numpy(t).to('cuda:0')).sum().backward()
    m.raise_if_overflow()
    assert rel(xg.grad.cpu().numpy(), dx_ref) < tol
    named = dict(m.named_parameters())
    assert all(p.grad is not None for p in named.values())
    for n, gr in grads_ref.items():
        assert rel(named[n].grad.cpu().numpy(), gr) < tol, n


def _sgb_case(name):
    from stofnet_amd.stofnet import SemiGlobalBlock
    var, g = SGB_VARIANTS[name], golden('f16_sgb_variants')
    seed = int(g[f'{name}.seed'])
    blk = SemiGlobalBlock(*var['ctor'])
    params = variant_params({n: tuple(t.shape) for n, t in blk.state_dict().items()}, seed)
    return var, blk, params, sgb_input(var['N'], var['ctor'][0], var['L'], seed), g[f'{name}.y']


@pytest.mark.parametrize('name', list(SGB_VARIANTS))
def test_oracle_semi_global_block_variants_match_reference(name):
    var, blk, params, x, y_ref = _sgb_case(name)
    y = so.semi_global_block(torch.from_numpy(x), params, '', var['ctor'][2], torch.float32)
    assert rel(y.numpy(), y_ref) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(SGB_VARIANTS))
def test_gpu_semi_global_block_variants_match_reference(name):
    """SemiGlobalBlock(in, out, sample_scale, kernel_size).forward standalone (models/stofnet.py:98-117) for other widths,
    scales and kernel sizes: channel-last MFMA convolutions + pool + generic-width upsample-add, exact-fp32 mode."""
    var, blk, params, x, y_ref = _sgb_case(name)
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    blk = blk.to('cuda:0')
    y = blk(torch.from_numpy(x).to('cuda:0'))
    assert y.shape == y_ref.shape
    assert rel(y.cpu().numpy(), y_ref) < 1e-5
    with pytest.raises(RuntimeError):              # odd remainder (Q1) / fewer samples than a window, as in the reference
        blk(torch.from_numpy(x[..., :var['ctor'][2] + 1 if var['ctor'][2] > 2 else 1]).to('cuda:0'))
