"""Importable alias for the ``st-nerf_amd/`` package directory.

The product tree lives in ``st-nerf_amd/`` (the name the build contract fixes); a hyphen is
not importable, so this shim maps ``import stnerf_amd.<x>`` onto the files in that directory.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_pkg = _os.path.join(_os.path.dirname(_here), "st-nerf_amd")
__path__.insert(0, _pkg)
with open(_os.path.join(_pkg, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg, "__init__.py"), "exec"))
