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
        EDSR_1D(num_channels=1, num_features=64, num_blocks=8, upscale_factor=4)


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
