"""MI355X-native CTC-attention mispronunciation-detection hot path.

Drop-in host layer (reference-shaped Python classes) over ``libmdd_hip.so`` -- a C-ABI
library of hand-written gfx950 HIP kernels (see include/mdd_hip.h)."""
__version__ = "0.1.0"
