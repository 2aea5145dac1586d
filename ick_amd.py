"""Import shim: the package directory is named after the reference repo
(`image-captioning-with-external-knowledge_amd/`), which is not a valid Python identifier.
`import ick_amd` loads that directory as the package `ick_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image-captioning-with-external-knowledge_amd")
_spec = importlib.util.spec_from_file_location(
    "ick_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ick_amd"] = _mod
_spec.loader.exec_module(_mod)
