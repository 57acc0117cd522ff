"""Import shim: the package directory is named `rs-tfhe_amd/` (not a Python
identifier), so `import rs_tfhe_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rs-tfhe_amd")
_spec = importlib.util.spec_from_file_location(
    "rs_tfhe_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rs_tfhe_amd"] = _mod
_spec.loader.exec_module(_mod)
