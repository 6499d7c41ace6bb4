"""Import shim: the package directory is named ``dualvgr-videoqa_amd`` (not a valid Python
identifier), so ``import dualvgr_videoqa_amd`` lands here and loads that directory as the package
``dualvgr_videoqa_amd`` (sub-modules import normally afterwards)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "dualvgr-videoqa_amd")
_spec = _ilu.spec_from_file_location(
    "dualvgr_videoqa_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["dualvgr_videoqa_amd"] = _mod
_spec.loader.exec_module(_mod)
