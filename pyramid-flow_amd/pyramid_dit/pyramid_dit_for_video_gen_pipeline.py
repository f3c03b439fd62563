"""Module path of the reference's pipeline class (pyramid_dit/pyramid_dit_for_video_gen_pipeline.py:114)."""
from pyflow_hip.pipeline import PyramidDiTForVideoGeneration  # noqa: F401
