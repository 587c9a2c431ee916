"""The reference's import lines resolve against this repo unchanged (main.py:18-22 of hahnec/stofnet):
    from models import StofNet, ZonziniNetLarge, ..., GradPeak, ...
    from utils.mask2samples import coords2mask, mask2nested_list, mask2coords
    from utils.gaussian import gaussian_kernel
    from utils.hilbert import hilbert_transform
    from utils.metrics import toa_rmse
CPU only: importing must not need the GPU or the built library; constructing StofNet must not either."""
import numpy as np
import pytest


def test_reference_import_lines_resolve():
    from models import StofNet, ZonziniNetLarge, ZonziniNetSmall, SincNet, GradPeak, Kuleshov, EDSR_1D, ESPCN_1D, WaveUnet  # noqa: F401
    from utils.mask2samples import coords2mask, mask2nested_list, mask2coords  # noqa: F401
    from utils.gaussian import gaussian_kernel
    from utils.hilbert import hilbert_transform  # noqa: F401
    from utils.metrics import toa_rmse  # noqa: F401
    from utils.sample_shuffle import SampleShuffle1D  # noqa: F401
    from models.stofnet import StofNet as S2, SemiGlobalBlock  # noqa: F401
    from models.gradpeak import GradPeak as G2, toa_detect, grad_peak_detect  # noqa: F401
    import stofnet_amd
    assert StofNet is stofnet_amd.StofNet is S2 and GradPeak is stofnet_amd.GradPeak is G2
    assert np.allclose(gaussian_kernel(7, 1)[:3], [0.004433, 0.054006, 0.242036], atol=1e-6)
    m = StofNet(upsample_factor=4)
    assert len(m.state_dict()) == 30 and sum(p.numel() for p in m.parameters()) == 645764
    assert len(StofNet(4, semi_global_scale=1).state_dict()) == 26
    with pytest.raises(NotImplementedError, match='out of scope'):
        WaveUnet()
    # the two baselines that ride on SampleShuffle1D are built (main.py:139-142 constructor calls)
    e = EDSR_1D(num_channels=1, num_features=64, num_blocks=8, upscale_factor=4)
    assert len(e.state_dict()) == 38 and e.conv_output.weight.shape == (1, 16, 3)
    assert ESPCN_1D(upscale_factor=4).conv3.weight.shape == (4, 32, 3)


def test_shuffle_riders_have_the_reference_parameter_names():
    """State-dict names and shapes of EDSR_1D / ESPCN_1D equal those of the reference's modules (fixture
    tests/golden/f11_shuffle_riders.npz holds the reference's parameters by name), so its checkpoints load strictly."""
    import os
    import torch
    from conftest import ROOT
    from models import EDSR_1D, ESPCN_1D
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'f11_shuffle_riders.npz'))
    cases = {'edsr_r4': EDSR_1D(num_channels=1, num_features=16, num_blocks=2, upscale_factor=4),
             'edsr_r2': EDSR_1D(num_channels=1, num_features=8, num_blocks=1, upscale_factor=2),
             'espcn_r4': ESPCN_1D(upscale_factor=4), 'espcn_r10': ESPCN_1D(upscale_factor=10)}
    for tag, model in cases.items():
        ref = {k.split('__p__')[1]: g[k] for k in g.files if k.startswith(tag + '__p__')}
        sd = model.state_dict()
        assert sorted(sd) == sorted(ref), tag
        assert all(tuple(sd[k].shape) == ref[k].shape for k in sd), tag
        model.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)
    # ESPCN's initialisation (models/espcn_1d.py:18-29): zero biases, std 0.001 for the layer fed by 32 channels
    torch.manual_seed(0)
    m = ESPCN_1D(upscale_factor=4)
    assert float(m.conv1.bias.detach().abs().max()) == 0.0 and 5e-4 < float(m.conv3.weight.detach().std()) < 2e-3
    assert abs(float(m.conv2.weight.detach().std()) - (2.0 / (32 * 3)) ** 0.5) < 0.02


def test_config_keeps_the_reference_keys():
    import os
    from conftest import ROOT
    from stofnet_amd import config as cm
    cfg = cm.load(os.path.join(ROOT, 'config.yaml'))
    for key in ('seed', 'logging', 'device', 'model', 'model_file', 'batch_size', 'lr', 'epochs', 'weight_decay',
                'upsample_factor', 'evaluate', 'patience', 'delta', 'lambda_value', 'mask_amplitude', 'kernel_size', 'sigma',
                'th', 'nms_win_size', 'sequences', 'rf_scale_factor', 'ch_gap', 'clutter_db', 'temporal_filter',
                'pow_law_opt', 'angle_threshold', 'etol', 'crop_ratio', 'snr_db', 'data_path', 'ubx_dir', 'loc_dir',
                'map_dir', 'data_dir'):
        assert key in cfg, key
    assert cfg.loc_dir.endswith('03_PALA/PALA_data_InSilicoFlow/') and cfg.map_dir.endswith('chris/PALA_data_InSilicoFlow/')
