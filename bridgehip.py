"""Import shim: the product package lives in the directory `bridge.jl_amd/` (not a valid Python
identifier), so `import bridgehip` loads that directory as the package `bridgehip`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bridge.jl_amd")
_spec = importlib.util.spec_from_file_location("bridgehip", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bridgehip"] = _mod
_spec.loader.exec_module(_mod)
