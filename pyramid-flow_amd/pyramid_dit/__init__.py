"""Drop-in for the reference's `pyramid_dit` package (pyramid_dit/__init__.py:1-3)."""
from pyflow_hip.pipeline import PyramidDiTForVideoGeneration  # noqa: F401
from pyflow_hip.flux import FluxEngine as PyramidFluxTransformer  # noqa: F401
