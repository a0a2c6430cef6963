"""The reference's ``layers`` package for the render path (layers/__init__.py:1-3)."""
from .RaySamplePoint import RaySamplePoint, intersection  # noqa: F401
from .render_layer import VolumeRenderer, gen_weight  # noqa: F401
