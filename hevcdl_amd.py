"""Import shim: the package directory name `hevc-deep-learning-pipeline_amd` is not a valid Python identifier,
so `import hevcdl_amd` loads it from its directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hevc-deep-learning-pipeline_amd")
_spec = importlib.util.spec_from_file_location("hevcdl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hevcdl_amd"] = _mod
_spec.loader.exec_module(_mod)
