"""utils/sample_shuffle.py of the reference on the gfx950 kernel."""
from stofnet_amd.sample_shuffle import SampleShuffle1D, sample_shuffle  # noqa: F401
