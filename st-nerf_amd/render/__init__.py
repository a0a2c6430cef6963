"""Mirror of the reference's ``render`` package for the layered renderer (render/__init__.py:5)."""
from stnerf_amd.path_renderer import LayeredNeuralRenderer  # noqa: F401
