"""CPU tests of the host-side mirror: config loader, metrics, entry-point plumbing."""
import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from stofnet_amd import config as cm
from stofnet_amd.metrics import toa_rmse


def test_config_load_merge_types():
    cfg = cm.merge(cm.load(f'{ROOT}/config.yaml'),
                   cm.from_cli(['th=Null', 'lr=1e-3', 'model=gradpeak', 'sequences=[1,2]', 'evaluate=True',
                                'model_file=different-armadillo_x', 'th2=.015']))
    assert cfg.th is None and cfg.lr == 1e-3 and cfg.model == 'gradpeak' and cfg.sequences == [1, 2]
    assert cfg.evaluate is True and cfg.th2 == 0.015 and cfg.weight_decay == 1e-8 and cfg.seed == 3008
    assert cfg.upsample_factor == 4 and cfg.nms_win_size == 20 and cfg.rf_scale_factor == 10
    assert cfg.ubx_dir.endswith('chris/PALA_data_InSilicoFlow/')       # ${data_path} interpolation
    cfg.fs = 1.5                                                         # runtime-added keys (main.py:72-74)
    assert cfg.fs == 1.5
    with pytest.raises(ValueError):
        cm.from_cli(['novalue'])


def test_toa_rmse_refuses_cpu_tensors():
    """No CPU fallback in the product: the device metric raises for host tensors (the CPU restatement is
    oracle/pickers_oracle.py:toa_rmse, pinned in test_oracle_golden.py)."""
    g = golden('f7_metrics')
    with pytest.raises(RuntimeError, match='ROCm device only'):
        toa_rmse(torch.from_numpy(g['gt']), torch.from_numpy(g['es']), tol=1)
