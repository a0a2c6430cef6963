"""The one piece of the reference's ``data`` package the render path needs: checkpoint discovery
(data/datasets/utils.py:42-60, imported as ``from data import get_iteration_path`` by
render/layered_neural_renderer.py:6).  Datasets and image I/O are out of scope."""
from stnerf_amd.render.checkpoint import get_iteration_path  # noqa: F401
