"""Import shim: `import legenddsp_jl_amd` loads the package that lives in the
directory `legenddsp.jl_amd/` (a dot is not legal in a Python package name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "legenddsp.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "legenddsp_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["legenddsp_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
