"""Import shim: the package directory is named ``pytorch-glow_amd`` (not a valid Python identifier), so
``import pytorch_glow_amd`` loads that directory as the package of this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pytorch-glow_amd")
_spec = importlib.util.spec_from_file_location("pytorch_glow_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pytorch_glow_amd"] = _mod
_spec.loader.exec_module(_mod)
