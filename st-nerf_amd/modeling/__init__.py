"""The reference's ``modeling`` package for the render path (modeling/__init__.py:3-7)."""
from .layered_rfrender import LayeredRFRender
from .motion_net import MotionNet  # noqa: F401
from .spacenet import SpaceNet  # noqa: F401


def build_layered_model(cfg, camera_num=0, scale=None, shift=None):
    """modeling/__init__.py:5."""
    return LayeredRFRender(cfg, camera_num=camera_num, scale=scale, shift=shift)
