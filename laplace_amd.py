"""Import alias for the package in ./laplace-gnn-recommendation_amd/.

The directory name carries the upstream repository's name, whose hyphens are not a valid
Python identifier; `import laplace_amd` loads that directory as the package `laplace_amd`
(sub-modules import normally: `laplace_amd.model.lightgcn`, `laplace_amd.config`, ...).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "laplace-gnn-recommendation_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
