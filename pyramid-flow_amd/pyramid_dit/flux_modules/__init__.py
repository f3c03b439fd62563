"""Drop-in for pyramid_dit/flux_modules/__init__.py:1-3 (miniFLUX): the transformer, its two block operators and the
prompt-encoder wrapper, all running on the HIP engine."""
from pyflow_hip.flux import PyramidFluxTransformer  # noqa: F401
from pyflow_hip.blocks import FluxSingleTransformerBlock, FluxTransformerBlock  # noqa: F401
from pyflow_hip.text_encoder import FluxTextEncoderWithMask  # noqa: F401
