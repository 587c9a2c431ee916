"""Drop-in for the reference's `models` package (models/__init__.py): `from models import StofNet, ..., GradPeak`
(main.py:18) resolves unchanged.  StofNet and GradPeak run on the gfx950 kernels (stofnet_amd); EDSR_1D and
ESPCN_1D ride on the SampleShuffle1D kernel (their convolutions stay stock ATen, as in the reference); the other
comparison networks of the paper's table are outside the accelerated path and raise when constructed."""
from stofnet_amd import EDSR_1D, ESPCN_1D, GradPeak, StofNet  # noqa: F401
from stofnet_amd.stofnet import SemiGlobalBlock  # noqa: F401


def _out_of_scope(name):
    class _Baseline:
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(f'{name} is a comparison baseline of the reference and not part of the MI355X-native '
                                      f'StofNet path (SURVEY.md section 2: out of scope)')
    _Baseline.__name__ = _Baseline.__qualname__ = name
    return _Baseline


ZonziniNetLarge = _out_of_scope('ZonziniNetLarge')
ZonziniNetSmall = _out_of_scope('ZonziniNetSmall')
SincNet = _out_of_scope('SincNet')
Kuleshov = _out_of_scope('Kuleshov')
WaveUnet = _out_of_scope('WaveUnet')
