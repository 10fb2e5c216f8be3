"""Import alias: the package directory is named `msm-webgpu_amd/` (not a valid Python identifier), so
`import msm_webgpu_amd` loads it from there.  Nothing else lives in this file."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "msm-webgpu_amd")
_spec = importlib.util.spec_from_file_location(
    "msm_webgpu_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["msm_webgpu_amd"] = _mod
_spec.loader.exec_module(_mod)
