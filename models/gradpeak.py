"""models/gradpeak.py of the reference on the gfx950 kernels."""
from stofnet_amd.gradpeak import GradPeak, gaussian_kernel_1d, grad_peak_detect, toa_detect  # noqa: F401
