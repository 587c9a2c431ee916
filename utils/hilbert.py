"""utils/hilbert.py of the reference on the gfx950 kernels."""
from stofnet_amd.hilbert import HilbertTransform, hilbert_envelope, hilbert_transform  # noqa: F401
