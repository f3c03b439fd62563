"""Drop-in for the reference's `pyramid_dit` package (pyramid_dit/__init__.py:1-3): the same six names."""
from .pyramid_dit_for_video_gen_pipeline import PyramidDiTForVideoGeneration  # noqa: F401
from .flux_modules import FluxSingleTransformerBlock, FluxTransformerBlock, FluxTextEncoderWithMask  # noqa: F401
from .mmdit_modules import JointTransformerBlock, SD3TextEncoderWithMask  # noqa: F401
