"""Import shim: registers the package directory `kissabc.jl_amd/` (whose name
contains a dot and therefore cannot be imported directly) as `kissabc_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kissabc.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "kissabc_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kissabc_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
