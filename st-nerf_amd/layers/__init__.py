"""Mirror of the reference's ``layers`` package for the render path (layers/__init__.py:1-3)."""
from stnerf_amd.renderer import RaySamplePoint, VolumeRenderer, gen_weight, intersection  # noqa: F401
