"""demi_amd — MI355X-native schedule-space exploration engine behind DEMi's Scheduler / TestOracle
plugin surface.  The compute path is libdemi_gpu.so (hand-written gfx950 HIP kernels behind the C
ABI of include/demi_gpu.h); this package is the host-side mirror of the reference's interface."""
from . import types  # noqa: F401
