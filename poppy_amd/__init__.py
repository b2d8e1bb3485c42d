"""poppy_amd — MI355X-native morph hot path (feature-match -> dense-warp -> blend) of kallaballa/Poppy.

The product is libpoppy_hip.so (hand-written gfx950 HIP kernels + host C++ behind the C ABI of
include/poppy_hip.h).  This package only holds its build recipe, a thin ctypes binding used by the
tests / bench, and the integer-defined synthetic inputs.  There is no CPU fallback anywhere.
"""
from . import synth  # noqa: F401
