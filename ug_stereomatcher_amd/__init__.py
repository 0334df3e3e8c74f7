"""ug_stereomatcher_amd -- MI355X (gfx950) drop-in for ug_stereomatcher's disparity hot path.

The product is libugsm.so (HIP kernels + C++ runtime behind the C-ABI of include/ugsm.h).
This package is the thin host side: the ctypes binding, a mirror of the reference's
`MatchGPULib` class and of the `UG_matcher_gpu` node / `GetDisparitiesGPU.srv` boundary,
and the synthetic-pair generator the tests and bench.py share.  There is no CPU fallback.
"""
from ._lib import Context, UgsmError, level_dims, level_iterations, level_smooth_passes, threshold_schedule, \
    fovea_dims, pixel_iterations
from .match_gpu_lib import MatchGPULib

__all__ = ["Context", "UgsmError", "MatchGPULib", "level_dims", "level_iterations", "level_smooth_passes",
           "threshold_schedule", "fovea_dims", "pixel_iterations"]
