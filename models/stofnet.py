"""models/stofnet.py of the reference: StofNet, SemiGlobalBlock (gfx950 kernels)."""
from stofnet_amd.stofnet import SemiGlobalBlock, StofNet  # noqa: F401
