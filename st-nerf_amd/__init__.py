"""st-nerf_amd: MI355X-native layered-NeRF ray-march renderer (hot path of DarlingHang/st-nerf).

Sub-packages mirror the reference's module names for this path (``modeling``, ``layers``,
``utils``, ``config``) so that putting this directory on ``sys.path`` makes the reference's own
``from modeling import build_layered_model`` / ``from utils import layered_batchify_ray`` resolve
here.  All compute goes through the C-ABI library built from ``csrc/``.
"""
__version__ = "0.1.0"
