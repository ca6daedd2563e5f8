"""Import alias for the product package.

The package directory is `gan-2d-to-3d_amd/` (the name the project layout prescribes); a hyphen is
not importable, so `import gan2shape_amd` loads that directory as the package `gan2shape_amd`.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gan-2d-to-3d_amd")
_spec = importlib.util.spec_from_file_location(
    "gan2shape_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gan2shape_amd"] = _mod
_spec.loader.exec_module(_mod)
