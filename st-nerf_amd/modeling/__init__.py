"""Mirror of the reference's ``modeling`` package for the render path (modeling/__init__.py:3-7)."""
from stnerf_amd.renderer import LayeredRFRender, MotionNet, SpaceNet, build_layered_model  # noqa: F401
