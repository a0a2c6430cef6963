"""The reference's ``config`` package (config/__init__.py:7): ``from config import cfg``."""
from .defaults import CfgNode, _C as cfg, get_cfg_defaults  # noqa: F401
