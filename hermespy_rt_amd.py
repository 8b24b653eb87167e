"""Import shim: the product package lives in the directory `hermespy-rt_amd/` (a name Python
cannot import directly); `import hermespy_rt_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hermespy-rt_amd")
_spec = importlib.util.spec_from_file_location(
    "hermespy_rt_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hermespy_rt_amd"] = _mod
_spec.loader.exec_module(_mod)
