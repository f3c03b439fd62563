"""Drop-in for the reference's `video_vae` package (video_vae/__init__.py:1-3; inference classes only)."""
from pyflow_hip.vae import CausalVideoVAE  # noqa: F401
