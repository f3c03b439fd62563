"""Drop-in for the reference's `diffusion_schedulers` package (diffusion_schedulers/__init__.py:1)."""
from pyflow_hip.scheduler import PyramidFlowMatchEulerDiscreteScheduler  # noqa: F401
