"""utils/mask2samples.py of the reference on the gfx950 picker kernels."""
from stofnet_amd.mask2samples import (batch_mask2coords, coords2mask, get_amplitudes, get_maxima_positions,  # noqa: F401
                                      mask2coords, mask2nested_list, reduce_echoes)
