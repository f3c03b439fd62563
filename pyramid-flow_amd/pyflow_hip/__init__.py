"""pyflow_hip -- host side of the MI355X-native pyramidal flow-matching sampler.

Everything numerically heavy runs in libpyflow_hip.so (hand-written gfx950 HIP kernels) through
the C ABI of include/pyflow_hip.h; this package is the Python host that mirrors the reference's
classes (PyramidDiTForVideoGeneration, PyramidFluxTransformer, PyramidFlowMatchEulerDiscreteScheduler,
CausalVideoVAE).  torch is used for device memory, streams and torch.distributed only.
"""
from . import lib  # noqa: F401
