"""Import shim: the product package directory is ``3dsad-main_amd/`` (not a valid Python
identifier), so ``import sad_amd`` loads it under this name.  ``sad_amd.config``,
``sad_amd.ops`` ... then resolve to ``3dsad-main_amd/config.py``, ``3dsad-main_amd/ops.py`` ..."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "3dsad-main_amd")
_spec = importlib.util.spec_from_file_location(
    "sad_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sad_amd"] = _mod
_spec.loader.exec_module(_mod)
