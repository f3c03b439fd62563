"""Drop-in for pyramid_dit/mmdit_modules/__init__.py:1-3 (SD3-style MMDiT)."""
from pyflow_hip.flux import PyramidDiffusionMMDiT  # noqa: F401
from pyflow_hip.blocks import JointTransformerBlock  # noqa: F401
from pyflow_hip.text_encoder import SD3TextEncoderWithMask  # noqa: F401
