"""utils/gaussian.py of the reference (host helper of the training loss)."""
from stofnet_amd.training import gaussian_kernel  # noqa: F401
