"""Import alias for the package directory ``robust-pose-estimator_amd/`` (a hyphen is not a valid Python
identifier).  ``import rpe_amd`` loads that directory as the package ``rpe_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'robust-pose-estimator_amd')
_spec = importlib.util.spec_from_file_location('rpe_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['rpe_amd'] = _mod
_spec.loader.exec_module(_mod)
