"""Importable alias for the ``st-nerf_amd/`` package directory.

The product tree lives in ``st-nerf_amd/`` (the name the build contract fixes); a hyphen is not importable, so this package
puts that directory first on its ``__path__`` -- ``import stnerf_amd.<x>`` then resolves to ``st-nerf_amd/<x>`` through the
normal import machinery (no ``exec``) -- and re-exports what ``st-nerf_amd/__init__.py`` defines (docstring, version).
"""
import importlib.util as _util
import os as _os

_pkg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "st-nerf_amd")
if not _os.path.isdir(_pkg):
    raise ImportError(f"{_pkg} is missing: the stnerf_amd alias must sit beside the st-nerf_amd/ package directory")
__path__.insert(0, _pkg)
_spec = _util.spec_from_file_location(__name__ + "._package_init", _os.path.join(_pkg, "__init__.py"))
_init = _util.module_from_spec(_spec)
_spec.loader.exec_module(_init)
__doc__ = _init.__doc__
__version__ = _init.__version__
