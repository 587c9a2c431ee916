"""MI355X-native StofNet inference path (drop-in for the reference's
`from models import StofNet, GradPeak`, `utils.sample_shuffle`, `utils.hilbert`,
`utils.mask2samples`).  All compute goes through libstofnet_amd.so."""
from .stofnet import StofNet, SemiGlobalBlock          # noqa: F401
from .sample_shuffle import SampleShuffle1D            # noqa: F401
from .mask2samples import mask2coords, get_maxima_positions, coords2mask  # noqa: F401
from .hilbert import hilbert_transform, HilbertTransform  # noqa: F401
from .gradpeak import GradPeak, toa_detect, grad_peak_detect  # noqa: F401
from .baselines import EDSR_1D, ESPCN_1D               # noqa: F401
