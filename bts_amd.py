"""Import shim: the product package directory is named `3d-brain-tumor-segmentation_amd` (not a valid Python
identifier), so `import bts_amd` loads it under this alias, sub-packages included."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), '3d-brain-tumor-segmentation_amd')
_spec = importlib.util.spec_from_file_location('bts_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['bts_amd'] = _mod
_spec.loader.exec_module(_mod)
