"""The reference's ``render`` package for the layered renderer (render/__init__.py:5)."""
from .checkpoint import get_iteration_path, load_reference_checkpoint  # noqa: F401
from .layered_neural_renderer import LayeredNeuralRenderer  # noqa: F401
from .render_pose import render_pose, to_uint8  # noqa: F401
