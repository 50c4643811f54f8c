"""Import shim: makes the package directory ``online-neural-cdes_amd/`` importable as ``ncde_amd``
(a hyphen is not legal in a Python module name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "online-neural-cdes_amd")
_spec = importlib.util.spec_from_file_location(
    "ncde_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ncde_amd"] = _mod
_spec.loader.exec_module(_mod)
