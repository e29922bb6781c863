"""Import shim: the package directory is `see-vcn_amd/` (not a valid identifier), so
`import seevcn_amd` loads that directory as the package `seevcn_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "see-vcn_amd")
_spec = importlib.util.spec_from_file_location(
    "seevcn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["seevcn_amd"] = _mod
_spec.loader.exec_module(_mod)
