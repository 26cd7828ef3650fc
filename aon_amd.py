"""Import shim: exposes the package directory ``articulated-object-nerf_amd/`` (whose
name is not a valid Python identifier) under the importable name ``aon_amd``.

    import aon_amd                       # -> the package in articulated-object-nerf_amd/
    from aon_amd.models.vanilla_nerf.model import NeRF
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "articulated-object-nerf_amd")
_spec = importlib.util.spec_from_file_location(
    "aon_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["aon_amd"] = _mod
_spec.loader.exec_module(_mod)
