"""Import shim: the package directory is `time-r1_amd/` (not a valid Python identifier); `import time_r1_amd` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "time-r1_amd")
_spec = importlib.util.spec_from_file_location("time_r1_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["time_r1_amd"] = _mod
_spec.loader.exec_module(_mod)
