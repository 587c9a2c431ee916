"""utils/metrics.py of the reference on the gfx950 kernel."""
from stofnet_amd.metrics import toa_rmse  # noqa: F401
